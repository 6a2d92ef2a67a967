#!/bin/bash
# r05y: the projection shortcut's BatchNorm applied inside the block output's pass (lp_bn_apply_seg_rbn) against a pass of its own that writes the normalised
# shortcut and is read back (LP_BN_APPLY_RBN=0): tests on the device, step A/B (alternating processes)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_step_parity.py -q -m gpu -k "projection_shortcut or blockwise or c2full or c1 or c5" -x 2>&1 | tail -3 | tee gpurun_out/r05y_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_BN_APPLY_RBN=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_APPLY_RBN=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05y_step_ab.txt
  done
done
