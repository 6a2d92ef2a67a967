# after the interface-hardening commits: device suite twice, smoke, default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do (timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r02_final_pytest_gpu_run$i.log; tail -1 gpurun_out/r02_final_pytest_gpu_run$i.log; done
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2) > gpurun_out/r02_final_smoke.log; tail -1 gpurun_out/r02_final_smoke.log
timeout 600 python bench.py > gpurun_out/r02_final_bench_n1.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_n1.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02_final_bench_n1_noprofile.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_n1_noprofile.json.log | cut -c1-200
