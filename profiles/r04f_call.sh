# Round 4, call F: tightened parity bars + c4full + HALO-vs-fp32 + the new bench lines on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ls oracle/_ref 2>&1 | head -3
(timeout 1500 python -m pytest tests/test_step_parity.py tests/test_gpu_fullsize.py tests/test_segmented_bn.py tests/test_differential_vs_reference.py -q -m gpu --timeout 900 -p no:cacheprovider -rs -s 2>&1 | grep -v "^$" | tail -60) > gpurun_out/r04f_pytest_parity.log; tail -30 gpurun_out/r04f_pytest_parity.log
timeout 900 python profiles/parity_report.py c2full c4full > gpurun_out/r04f_parity_device.jsonl 2> gpurun_out/r04f_parity.err; tail -2 gpurun_out/r04f_parity.err
timeout 900 python bench.py > gpurun_out/r04f_bench_n1.json.log 2>&1; tail -1 gpurun_out/r04f_bench_n1.json.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('attainable_frac'), d['roofline'].get('traffic_source'))
print(d.get('cpu_baseline'))
for k,v in d.get('secondary',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('decode_prune'), v.get('error'))
"
