# round 2, call 3: fp32 validation path + peaked-heat-map parity at the BASELINE configs on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python profiles/parity_report.py 2>&1 | grep "^{") > gpurun_out/r02c_parity.jsonl; wc -l gpurun_out/r02c_parity.jsonl
(timeout 900 python -m pytest tests/test_fp32_parity.py tests/test_step_parity.py -q -m gpu --timeout 400 -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/r02c_pytest_gpu.log; tail -5 gpurun_out/r02c_pytest_gpu.log
