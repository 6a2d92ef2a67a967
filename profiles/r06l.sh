#!/bin/bash
# r06l: which of the two GELU fusions pays: none / backward only / both, alternating processes in one call (ViT-S/16 step, BASELINE C4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  for m in 0 bwd 1; do
    LP_VIT_GELU_FUSED=$m timeout 300 python bench.py --backbone vits_dino --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VIT_GELU_FUSED=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06l_vit_step_ab.txt
  done
done
