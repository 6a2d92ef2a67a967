# Round-end validation on one MI355X box (gpurun): GPU test suite, smoke(), the driver's default bench line, the C4 line, a functional
# 2-rank run of the data-parallel path on ONE GPU (gloo, both ranks on device 0: exercises SyncBatchNorm / bucketed all-reduce), and
# the rocprofv3 kernel traces the roofline entry is checked against (default command, and the serialised run = what the HIP events see).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/pytest_gpu_final.log
tail -2 gpurun_out/pytest_gpu_final.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/smoke_final.log
tail -1 gpurun_out/smoke_final.log
timeout 400 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-260
timeout 300 python bench.py --backbone vits_dino > gpurun_out/bench_vit_final.log 2>&1; tail -1 gpurun_out/bench_vit_final.log | cut -c1-200
for bb in resnet50 vits_dino; do
  LP_FORCE_DEVICE=0 LP_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 2 --steps 2 --warmup 1 --labeled 16 --unlabeled 32 --backbone $bb --no-cpu-baseline > gpurun_out/bench_2rank_gloo_${bb}_final.log 2>&1
  tail -1 gpurun_out/bench_2rank_gloo_${bb}_final.log | cut -c1-200
done
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o final -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_final.log 2>&1; tail -1 gpurun_out/prof_final.log | cut -c1-120
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final_serial -o serial -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_final_serial.log 2>&1; tail -1 gpurun_out/prof_final_serial.log | cut -c1-120
