cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python profiles/parity_report.py 2>&1 | grep "^{") > gpurun_out/r02j_parity.jsonl; wc -l gpurun_out/r02j_parity.jsonl
(timeout 600 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 400 -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED" | grep -v "where\|built-in" | cut -c1-400) > gpurun_out/r02j_pytest.log; tail -6 gpurun_out/r02j_pytest.log
