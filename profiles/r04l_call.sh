# round 4, call l: fixed-point (lp_fxsum) BatchNorm sums vs the slot-row build of commit c9a980d (build/ab_slots), 3 A/B pairs in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04l_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04l_bench_fx_$i.json.log
  echo "pair $i: slots $(grep -o '"value": [0-9.]*' gpurun_out/r04l_bench_slots_$i.json.log | head -1)  fx $(grep -o '"value": [0-9.]*' gpurun_out/r04l_bench_fx_$i.json.log | head -1)"
done
