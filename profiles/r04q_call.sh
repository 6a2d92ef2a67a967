# round 4, call q: parameter gradients spread over the grid (one thread per channel) + BatchNorm walks on one resident round (1280) vs slot rows;
# also the cap at 1536 / 1280 for the record, kernel trace, and the device suite on the final kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04q_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04q_bench_fx_$i.json.log
  LP_BN_GRID_CAP=2048 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04q_bench_cap2048_$i.json.log
  LP_BN_GRID_CAP=1536 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04q_bench_cap1536_$i.json.log
  echo "pair $i: slots $(v gpurun_out/r04q_bench_slots_$i.json.log) fx(1280) $(v gpurun_out/r04q_bench_fx_$i.json.log) cap2048 $(v gpurun_out/r04q_bench_cap2048_$i.json.log) cap1536 $(v gpurun_out/r04q_bench_cap1536_$i.json.log)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04q -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04q/t_results.db > gpurun_out/r04q_kernel_stats_fx.txt 2>&1
grep -h "bn_bwd_apply\|lp::bn_apply" gpurun_out/r04q_kernel_stats_fx.txt | cut -c1-40,105-175
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r04q_pytest_gpu.log; tail -2 gpurun_out/r04q_pytest_gpu.log
