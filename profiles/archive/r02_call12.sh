cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r02l_pytest_gpu.log; tail -3 gpurun_out/r02l_pytest_gpu.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02l_bench_$i.json.log 2>&1; tail -1 gpurun_out/r02l_bench_$i.json.log | cut -c1-330; done
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --size 256 > gpurun_out/r02l_bench_256.json.log 2>&1; tail -1 gpurun_out/r02l_bench_256.json.log | cut -c1-330
