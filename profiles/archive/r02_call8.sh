# round 2, call 8: 2-rank functional run of the data-parallel path on ONE GPU (gloo, both ranks on device 0: SyncBatchNorm with two segments,
# gradient buckets leaving during backward, packed scalar means), then the device suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for bb in resnet50 vits_dino; do
  LP_FORCE_DEVICE=0 LP_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 2 --steps 2 --warmup 1 --labeled 16 --unlabeled 32 --backbone $bb --no-cpu-baseline > gpurun_out/r02h_bench_2rank_gloo_${bb}.log 2>&1
  tail -1 gpurun_out/r02h_bench_2rank_gloo_${bb}.log | cut -c1-600
done
(timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r02h_pytest_gpu.log; tail -3 gpurun_out/r02h_pytest_gpu.log
