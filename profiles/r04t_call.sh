# round 4, call t: 2-rank data-parallel run on ONE GPU (gloo transport, both ranks on device 0): a functional check of SyncBatchNorm on the
# fixed-point sums (int64 all-reduce / one-shot gather) and the bucketed gradient all-reduce with the real kernels - not a scaling number
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LP_FORCE_DEVICE=0 LP_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --labeled 8 --unlabeled 16 > gpurun_out/r04t_bench_2rank_gloo_allreduce.log 2>&1; tail -1 gpurun_out/r04t_bench_2rank_gloo_allreduce.log | cut -c1-900
LP_FORCE_DEVICE=0 LP_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --labeled 8 --unlabeled 16 --syncbn-gather > gpurun_out/r04t_bench_2rank_gloo_gather.log 2>&1; tail -1 gpurun_out/r04t_bench_2rank_gloo_gather.log | cut -c1-900
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --labeled 8 --unlabeled 16 > gpurun_out/r04t_bench_1rank.log 2>&1; tail -1 gpurun_out/r04t_bench_1rank.log | cut -c1-400
