# round 4, call p: bn_bwd_apply conversion 8 deep; grid of the BatchNorm walks capped at 2048 (as is) / 1280 (one resident round) / 1024; vs slot rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04p_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04p_bench_fx_$i.json.log
  LP_BN_GRID_CAP=1280 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04p_bench_cap1280_$i.json.log
  LP_BN_GRID_CAP=1024 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04p_bench_cap1024_$i.json.log
  echo "pair $i: slots $(v gpurun_out/r04p_bench_slots_$i.json.log) fx $(v gpurun_out/r04p_bench_fx_$i.json.log) cap1280 $(v gpurun_out/r04p_bench_cap1280_$i.json.log) cap1024 $(v gpurun_out/r04p_bench_cap1024_$i.json.log)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
for cap in 2048 1280; do
LP_BN_GRID_CAP=$cap LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04p_$cap -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04p_$cap/t_results.db > gpurun_out/r04p_kernel_stats_cap$cap.txt 2>&1
done
grep -h "bn_bwd_apply\|lp::bn_apply" gpurun_out/r04p_kernel_stats_cap2048.txt gpurun_out/r04p_kernel_stats_cap1280.txt | cut -c1-40,105-175
