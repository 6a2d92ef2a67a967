cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python profiles/decode_microbench.py > gpurun_out/r02k_decode_microbench.jsonl 2>&1; tail -5 gpurun_out/r02k_decode_microbench.jsonl
(timeout 600 python -m pytest tests/test_emu_decode.py tests/test_reference_cases.py -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r02k_pytest.log; tail -2 gpurun_out/r02k_pytest.log
