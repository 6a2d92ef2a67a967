# round 2, call 6: the whole device suite + smoke + the driver's default bench command (headline + secondary lines)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r02e_pytest_gpu.log; tail -4 gpurun_out/r02e_pytest_gpu.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2) > gpurun_out/r02e_smoke.log; tail -1 gpurun_out/r02e_smoke.log
(time timeout 600 python bench.py) > gpurun_out/r02e_bench_default.json.log 2>&1; tail -5 gpurun_out/r02e_bench_default.json.log | cut -c1-400
