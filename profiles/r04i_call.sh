# Round 4, call I: slot rows (bit-reproducible, no wait at the zeroing, backward reduction outside the convolution's bracket) vs LP_STATS_ATOMIC=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 12 > gpurun_out/r04i_bench_slots_$i.json.log 2>&1
  LP_STATS_ATOMIC=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 12 > gpurun_out/r04i_bench_atomic_$i.json.log 2>&1
done
for f in gpurun_out/r04i_bench_*_?.json.log; do echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 > gpurun_out/r04i_bench_slots_prof.json.log 2>&1
LP_STATS_ATOMIC=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 > gpurun_out/r04i_bench_atomic_prof.json.log 2>&1
for t in slots atomic; do echo $t $(tail -1 gpurun_out/r04i_bench_${t}_prof.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['mfma_ms_per_step'], r['frac'], r['attainable_frac'])"); done
(timeout 900 python -m pytest tests/test_step_parity.py tests/test_segmented_bn.py -q -m gpu --timeout 900 -p no:cacheprovider -rf 2>&1 | tail -5) > gpurun_out/r04i_pytest.log; tail -3 gpurun_out/r04i_pytest.log
