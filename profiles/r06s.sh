#!/bin/bash
# r06s: is Trainer.fit's "-6 %" per step or per fit() call?  The fit line at 6 / 12 / 24 / 48 timed steps next to the resident-batch headline on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident batch', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06s_fit_steps.txt
for n in 6 12 24 48; do
  timeout 400 python bench.py --fit --steps $n --warmup 3 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fit steps=$n', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06s_fit_steps.txt
done
