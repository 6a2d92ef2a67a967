#!/bin/bash
# r05q: lp_stem_wgrad_pool variants in the step: default build (2 workgroups per CU), 3 per CU (register cap 168: spills), 3 per CU with the per-channel
# constants in LDS; LP_STEM_WGRAD_POOL=0 as the reference.  Alternating processes, two rounds.
mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05q_pool_variants.txt; }
for i in 1 2; do
  LP_STEM_WGRAD_POOL=0 run "two-call form     "
  run "fused, default    "
  LP_HIP_LIB=$PWD/build/liblp_hip_nb3.so run "fused, 3 WGs/CU   "
  LP_HIP_LIB=$PWD/build/liblp_hip_nbc.so run "fused, 3 WGs, LDS consts"
  LP_HIP_LIB=$PWD/build/liblp_hip_nbc2.so run "fused, 2 WGs, LDS consts"
done
