# round 2, call 4: the step as one captured HIP graph
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_graph_step.py -q -m gpu --timeout 200 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r02d_pytest_graph.log; tail -4 gpurun_out/r02d_pytest_graph.log
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 > gpurun_out/r02d_bench_graph.json.log 2>&1; tail -2 gpurun_out/r02d_bench_graph.json.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 0 > gpurun_out/r02d_bench_eager.json.log 2>&1; tail -1 gpurun_out/r02d_bench_eager.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 --size 256 > gpurun_out/r02d_bench_graph_256.json.log 2>&1; tail -2 gpurun_out/r02d_bench_graph_256.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 --backbone vits_dino > gpurun_out/r02d_bench_graph_vit.json.log 2>&1; tail -2 gpurun_out/r02d_bench_graph_vit.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --graph 1 --views 4 --size 256 --labeled 16 --unlabeled 32 > gpurun_out/r02d_bench_graph_c5.json.log 2>&1; tail -2 gpurun_out/r02d_bench_graph_c5.json.log | cut -c1-300
