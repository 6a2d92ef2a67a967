# round 3, call 38: idle-gap analysis of the step (kernel trace), 2-rank functional run of the self-launching bench on one device (gloo)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace -d /tmp/r03_gap -o gap -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 3 > gpurun_out/r03ai_gap_bench.log 2>&1
python profiles/gap_analysis.py /tmp/r03_gap/gap_results.db > gpurun_out/r03ai_gap_analysis.txt 2>&1; head -70 gpurun_out/r03ai_gap_analysis.txt
LP_FORCE_DEVICE=0 LP_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --labeled 8 --unlabeled 16 > gpurun_out/r03ai_bench_2rank_gloo.log 2>&1; tail -3 gpurun_out/r03ai_bench_2rank_gloo.log | cut -c1-600
