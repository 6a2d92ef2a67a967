# round 2, call 4b: HIP-graph replay with and without the weight-gradient side stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LP_WGRAD_SIDE_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 > gpurun_out/r02d_bench_graph_noside.json.log 2>&1; tail -1 gpurun_out/r02d_bench_graph_noside.json.log | cut -c1-330
LP_WGRAD_SIDE_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 0 > gpurun_out/r02d_bench_eager_noside.json.log 2>&1; tail -1 gpurun_out/r02d_bench_eager_noside.json.log | cut -c1-330
LP_WGRAD_SIDE_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 --size 256 > gpurun_out/r02d_bench_graph_256_noside.json.log 2>&1; tail -1 gpurun_out/r02d_bench_graph_256_noside.json.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 0 --size 256 > gpurun_out/r02d_bench_eager_256.json.log 2>&1; tail -1 gpurun_out/r02d_bench_eager_256.json.log | cut -c1-330
