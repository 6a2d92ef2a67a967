# round 2, call 9: normalise-on-load for bn2 -> conv3 (a2 never materialised): device tests + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_step_parity.py tests/test_gpu_fullsize.py tests/test_emu_tracker.py -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r02i_pytest.log; tail -2 gpurun_out/r02i_pytest.log
for v in 1 0 1 0; do
  LP_NORM_ON_LOAD=$v timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02i_bench_norm$v.json.log 2>&1; echo "LP_NORM_ON_LOAD=$v"; tail -1 gpurun_out/r02i_bench_norm$v.json.log | cut -c1-160
done
LP_NORM_ON_LOAD=1 timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --size 256 > gpurun_out/r02i_bench_norm1_256.json.log 2>&1; tail -1 gpurun_out/r02i_bench_norm1_256.json.log | cut -c1-160
LP_NORM_ON_LOAD=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --size 256 > gpurun_out/r02i_bench_norm0_256.json.log 2>&1; tail -1 gpurun_out/r02i_bench_norm0_256.json.log | cut -c1-160
