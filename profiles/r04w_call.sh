# round 4, call w: grid cap of colreduce_kernel (the 18 un-fused BatchNorm reductions per step): 1024 (as committed) vs 2048 vs 512 workgroups
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04w_bench_1024_$i.json.log
  for c in 2048 512; do LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_cr$c.so timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04w_bench_${c}_$i.json.log; done
  echo "round $i: 1024 $(v gpurun_out/r04w_bench_1024_$i.json.log) 2048 $(v gpurun_out/r04w_bench_2048_$i.json.log) 512 $(v gpurun_out/r04w_bench_512_$i.json.log)"
done
for c in 2048 512; do
LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_cr$c.so LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04w_$c -o t -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04w_$c/t_results.db 2>&1 | grep "colreduce\|rows_reduce" | cut -c1-40,105-175
done
