# round 3, call 21: whole device suite + default bench on the tree with the HALO form, the Linear layers on the pipelined kernel, lp_loss_combine and the zero arena
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider -x 2>&1 | tail -6) > gpurun_out/r03t_pytest_gpu.log; tail -3 gpurun_out/r03t_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py > gpurun_out/r03t_bench_default.json.log 2>&1; tail -1 gpurun_out/r03t_bench_default.json.log | cut -c1-400
