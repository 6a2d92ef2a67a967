#!/bin/bash
# r05w: the projection shortcut's BatchNorm reductions taken by bn3's backward walk over the same gradient (lp_bn_bwd_apply_seg_ds) against a pass of
# their own (LP_BN_BWD_DS=0: lp_bn_bwd_reduce): tests on the device, step A/B (alternating processes)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_step_parity.py -q -m gpu -k "projection_shortcut or blockwise or c2full or c1" -x 2>&1 | tail -3 | tee gpurun_out/r05w_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_BN_BWD_DS=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_BWD_DS=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05w_step_ab.txt
  done
done
