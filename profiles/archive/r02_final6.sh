# the device suite of the closing tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider --tb=short > /tmp/suite1.log 2>&1; tail -1 /tmp/suite1.log; (grep -E "^(FAILED|E  )" /tmp/suite1.log | head -40; tail -3 /tmp/suite1.log) | cut -c1-300 > gpurun_out/r02_final_pytest_gpu_run3.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2) > gpurun_out/r02_final_smoke.log; tail -1 gpurun_out/r02_final_smoke.log
