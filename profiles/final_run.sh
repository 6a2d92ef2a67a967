cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/pytest_gpu_r1f.log
tail -2 gpurun_out/pytest_gpu_r1f.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/smoke_r1f.log
tail -1 gpurun_out/smoke_r1f.log
timeout 400 python bench.py > gpurun_out/bench_r1f.log 2>&1; tail -1 gpurun_out/bench_r1f.log | cut -c1-260
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r1f_20.log 2>&1; tail -1 gpurun_out/bench_r1f_20.log | cut -c1-200
timeout 300 python bench.py --backbone vits_dino --steps 20 --warmup 5 > gpurun_out/bench_vit_r1f.log 2>&1; tail -1 gpurun_out/bench_vit_r1f.log | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1f -o r1f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r1f.log 2>&1; tail -1 gpurun_out/prof_r1f.log | cut -c1-120
