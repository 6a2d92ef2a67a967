# round 4, call k: fixed-point (lp_fxsum) BatchNorm sums vs the slot-row build of commit c9a980d (build/ab_slots): device suite, then 3 A/B pairs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider -x 2>&1 | tail -25) > gpurun_out/r04k_pytest_gpu.log; tail -3 gpurun_out/r04k_pytest_gpu.log
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
for i in 1 2 3; do
  (cd build/ab_slots && timeout 300 python $B 2>&1 | tail -1) > gpurun_out/r04k_bench_slots_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04k_bench_fx_$i.json.log
  echo "pair $i: slots $(grep -o '"value": [0-9.]*' gpurun_out/r04k_bench_slots_$i.json.log | head -1)  fx $(grep -o '"value": [0-9.]*' gpurun_out/r04k_bench_fx_$i.json.log | head -1)"
done
