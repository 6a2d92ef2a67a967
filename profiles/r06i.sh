#!/bin/bash
# r06i: the optional "fp32 residual stream" policy (LP_RESIDUAL_FP32=1: block outputs as bf16 pairs, projection shortcuts added unrounded) on the device:
# kernel + engine tests, every logged scalar of the BASELINE fixtures under both policies, and what it costs in the step (alternating processes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3) | tee gpurun_out/r06i_pytest.txt
timeout 1500 python profiles/residual_fp32_report.py c1 c5 c5v4 c2 c2full 2>/dev/null | tee gpurun_out/r06i_residual_fp32_report.jsonl | cut -c1-400
for i in 1 2 3; do
  for m in 0 1; do
    LP_RESIDUAL_FP32=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESIDUAL_FP32=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06i_step_ab.txt
  done
done
