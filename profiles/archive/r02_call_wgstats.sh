# BatchNorm sums accumulated per persistent workgroup (atomics for every launch, no tile_stats_reduce) vs the workspace + reduce path: A/B in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -q -m gpu -k "emu_conv or ragged_shapes or segmented or emu_engine or step_parity or fullsize" --timeout 300 -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r02t_pytest.log; tail -1 gpurun_out/r02t_pytest.log
B="timeout 200 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  $B > gpurun_out/r02t_bench_wg_$i.json.log 2>&1; tail -1 gpurun_out/r02t_bench_wg_$i.json.log | cut -c80-200
  LP_STATS_ATOMIC_TILES=1152 $B > gpurun_out/r02t_bench_ws_$i.json.log 2>&1; tail -1 gpurun_out/r02t_bench_ws_$i.json.log | cut -c80-200
done
$B --size 256 > gpurun_out/r02t_bench_wg_256.json.log 2>&1; tail -1 gpurun_out/r02t_bench_wg_256.json.log | cut -c80-200
LP_STATS_ATOMIC_TILES=1152 $B --size 256 > gpurun_out/r02t_bench_ws_256.json.log 2>&1; tail -1 gpurun_out/r02t_bench_ws_256.json.log | cut -c80-200
