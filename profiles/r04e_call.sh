# Round 4, call E: device suite on the rebuilt library (slot-row sums, C % 8) + the measured parity of every step fixture, twice (bit-stability)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r04e_pytest_gpu.log; tail -5 gpurun_out/r04e_pytest_gpu.log
timeout 900 python profiles/parity_report.py c1 c2 c5 c5v4 c2full c4 > gpurun_out/r04e_parity_device_run1.jsonl 2> gpurun_out/r04e_parity_run1.err
timeout 900 python profiles/parity_report.py c2full c2 > gpurun_out/r04e_parity_device_run2.jsonl 2> gpurun_out/r04e_parity_run2.err
tail -2 gpurun_out/r04e_parity_run1.err
python - <<'PY'
import json
a=[json.loads(l) for l in open('gpurun_out/r04e_parity_device_run1.jsonl')]
b=[json.loads(l) for l in open('gpurun_out/r04e_parity_device_run2.jsonl')]
key=lambda r:(r['config'],r['precision'])
A={key(r):r for r in a}
for r in b:
    print(key(r), 'identical to run 1:', json.dumps(A[key(r)],sort_keys=True)==json.dumps(r,sort_keys=True))
PY
