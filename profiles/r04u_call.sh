# round 4, call u: weight-gradient launches handed to the side stream in groups (one main-stream event per LP_WGRAD_GROUP layers instead of per layer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  for gsz in 1 2 4 8; do LP_WGRAD_GROUP=$gsz timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04u_bench_g${gsz}_$i.json.log; done
  echo "round $i: g1 $(v gpurun_out/r04u_bench_g1_$i.json.log) g2 $(v gpurun_out/r04u_bench_g2_$i.json.log) g4 $(v gpurun_out/r04u_bench_g4_$i.json.log) g8 $(v gpurun_out/r04u_bench_g8_$i.json.log)"
done
for gsz in 1 4; do
LP_WGRAD_GROUP=$gsz timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04u_$gsz -o t -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > /dev/null 2>&1
python profiles/gap_analysis.py /tmp/r04u_$gsz/t_results.db > gpurun_out/r04u_gap_analysis_g$gsz.txt 2>&1; sed -n 2,9p gpurun_out/r04u_gap_analysis_g$gsz.txt
done
