# round 4, call o: bn_bwd_apply with segment 0's terms requested before the conversion barrier; 5 (one spill) vs 4 waves per SIMD; vs slot rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04o_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04o_bench_fx_$i.json.log
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_occ4.so timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04o_bench_occ4_$i.json.log
  echo "pair $i: slots $(grep -o '"value": [0-9.]*' gpurun_out/r04o_bench_slots_$i.json.log | head -1)  fx $(grep -o '"value": [0-9.]*' gpurun_out/r04o_bench_fx_$i.json.log | head -1) occ4 $(grep -o '"value": [0-9.]*' gpurun_out/r04o_bench_occ4_$i.json.log | head -1)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04o_fx -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04o_fx/t_results.db > gpurun_out/r04o_kernel_stats_fx.txt 2>&1
LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_occ4.so LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04o_occ4 -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04o_occ4/t_results.db > gpurun_out/r04o_kernel_stats_occ4.txt 2>&1
grep -h "bn_bwd_apply" gpurun_out/r04o_kernel_stats_fx.txt gpurun_out/r04o_kernel_stats_occ4.txt | cut -c1-40,105-175
