# round 6, the tree as it is handed over: whole device suite, smoke, the driver's default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r06_final4_pytest_gpu.log; tail -2 gpurun_out/r06_final4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160 | tee gpurun_out/r06_final4_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_final4_bench_n1.json.log 2>&1; tail -1 gpurun_out/r06_final4_bench_n1.json.log | cut -c1-330
