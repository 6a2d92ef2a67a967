# round 4, call r: what the fixed-point prologue of bn_bwd_apply costs - timing builds (wrong numerics): kt1 = zeros into LDS (no global loads, no
# conversion; LDS + barrier stay), kt2 = no LDS, no barrier (the kernel as it was with fp32 sums, at the new grid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04r_bench_fx_$i.json.log
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_kt1.so timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04r_bench_kt1_$i.json.log
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_kt2.so timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04r_bench_kt2_$i.json.log
  echo "pair $i: fx $(v gpurun_out/r04r_bench_fx_$i.json.log) kt1 $(v gpurun_out/r04r_bench_kt1_$i.json.log) kt2 $(v gpurun_out/r04r_bench_kt2_$i.json.log)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
for t in kt1 kt2; do
LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$t.so LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04r_$t -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04r_$t/t_results.db > gpurun_out/r04r_kernel_stats_$t.txt 2>&1
done
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04r_fx -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04r_fx/t_results.db > gpurun_out/r04r_kernel_stats_fx.txt 2>&1
grep -h "bn_bwd_apply" gpurun_out/r04r_kernel_stats_fx.txt gpurun_out/r04r_kernel_stats_kt1.txt gpurun_out/r04r_kernel_stats_kt2.txt | cut -c1-40,105-175
