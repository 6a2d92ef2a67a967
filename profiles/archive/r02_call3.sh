# round 2, call 3: fp32 validation path + peaked-heat-map parity at the BASELINE configs on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fp32_parity.py tests/test_step_parity.py -q -m gpu --timeout 400 -p no:cacheprovider -s 2>&1 | grep -E "PARITY|^E  |passed|failed|FAILED" | grep -v "where\|built-in" | cut -c1-700) > gpurun_out/r02c_pytest_gpu.log; tail -5 gpurun_out/r02c_pytest_gpu.log
