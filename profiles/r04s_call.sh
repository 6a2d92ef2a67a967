# round 4, call s: inference forward with layer1's 3x3 layers on conv_res2d_kernel's inference store pass (default) vs on conv_pipe_kernel
# (LP_CONV_RES2D=0), three alternating pairs; kernel trace of the predict step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --predict --no-cpu-baseline --no-profile --no-secondary --steps 20 --warmup 5"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  LP_CONV_RES2D=0 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04s_predict_pipe_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04s_predict_res2d_$i.json.log
  echo "pair $i: pipe $(v gpurun_out/r04s_predict_pipe_$i.json.log) res2d $(v gpurun_out/r04s_predict_res2d_$i.json.log)"
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04s -o t -- python bench.py --predict --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 2 > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04s/t_results.db > gpurun_out/r04s_predict_kernel_stats.txt 2>&1
head -8 gpurun_out/r04s_predict_kernel_stats.txt | cut -c1-60,105-160
