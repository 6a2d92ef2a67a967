# Round 4, call G: MALL probe (does reading a just-written tensor back to front hit the memory-side cache?) + re-run of the parity subset + bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe profiles/probe/mall_probe.hip 2> gpurun_out/r04g_mall_probe_build.log
timeout 120 /tmp/mall_probe > gpurun_out/r04g_mall_probe.txt 2>&1; cat gpurun_out/r04g_mall_probe.txt
(timeout 1500 python -m pytest tests/test_step_parity.py tests/test_gpu_fullsize.py tests/test_segmented_bn.py tests/test_widen_bench_helpers.py -q -m gpu --timeout 900 -p no:cacheprovider -rf 2>&1 | tail -25) > gpurun_out/r04g_pytest_parity.log; tail -12 gpurun_out/r04g_pytest_parity.log
timeout 600 python bench.py --fit --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r04g_bench_fit.json.log 2>&1; tail -1 gpurun_out/r04g_bench_fit.json.log | cut -c1-400
timeout 600 python bench.py --backbone vits_dino --no-cpu-baseline --no-secondary > gpurun_out/r04g_bench_vit.json.log 2>&1; tail -1 gpurun_out/r04g_bench_vit.json.log | cut -c1-300
