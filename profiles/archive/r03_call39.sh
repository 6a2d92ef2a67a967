# round 3, call 39: stream priorities - weight-gradient side stream at low priority (LP_SIDE_PRIORITY=1) / the step's stream at high priority
# (LP_SIDE_PRIORITY / LP_MAIN_PRIORITY were test switches of this run only: no effect, removed again - profiles/r03aj_priorities.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "
import ctypes as C, torch
torch.zeros(1, device='cuda')
hip = C.CDLL('libamdhip64.so'); lo, hi = C.c_int(0), C.c_int(0); print('rc', hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), 'least', lo.value, 'greatest', hi.value)
print('torch range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')
" > gpurun_out/r03aj_priorities.txt 2>&1
for rep in 1 2; do
  for v in "base" "LP_SIDE_PRIORITY=1" "LP_MAIN_PRIORITY=-1" "LP_WGRAD_SIDE_STREAM=0"; do
    env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v', d['value'], d['ms_per_step'])
    elif 'Error' in l or 'error' in l: print('$v', l.strip()[:200])"
  done
done >> gpurun_out/r03aj_priorities.txt 2>&1; cat gpurun_out/r03aj_priorities.txt
