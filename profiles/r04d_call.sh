# Round 4, call D: device suite on the slot-row (bit-reproducible) BatchNorm sums + A/B against the atomic form + the full default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r04d_pytest_gpu.log; tail -3 gpurun_out/r04d_pytest_gpu.log
for i in 1 2; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r04d_bench_slots_$i.json.log 2>&1
  LP_STATS_ATOMIC=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r04d_bench_atomic_$i.json.log 2>&1
done
for f in gpurun_out/r04d_bench_*_?.json.log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
timeout 900 python bench.py > gpurun_out/r04d_bench_n1.json.log 2>&1; tail -1 gpurun_out/r04d_bench_n1.json.log | cut -c1-600
