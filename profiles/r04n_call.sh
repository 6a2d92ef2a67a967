# round 4, call n: fixed-point sums, second version (correction terms of bn_bwd_apply converted once per workgroup into LDS, 5 waves per SIMD
# again; lp_bn_stats / lp_bn_bwd_reduce back on per-workgroup rows + an ordered reduction that emits fixed point) vs the slot-row build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_trunk_ops.py tests/test_emu_conv.py tests/test_emu_engine.py tests/test_gpu_fullsize.py tests/test_step_parity.py -q -m gpu --timeout 900 -p no:cacheprovider -x 2>&1 | tail -4) > gpurun_out/r04n_pytest_gpu.log; tail -2 gpurun_out/r04n_pytest_gpu.log
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04n_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04n_bench_fx_$i.json.log
  echo "pair $i: slots $(grep -o '"value": [0-9.]*' gpurun_out/r04n_bench_slots_$i.json.log | head -1)  fx $(grep -o '"value": [0-9.]*' gpurun_out/r04n_bench_fx_$i.json.log | head -1)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04n_fx -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04n_fx/t_results.db > gpurun_out/r04n_kernel_stats_fx.txt 2>&1
grep -h "colreduce\|bn_bwd_apply\|rows_reduce\|bn_finalize" gpurun_out/r04n_kernel_stats_fx.txt | cut -c1-40,105-175
