#!/bin/bash
# r05x: lp_bn_bwd_apply_seg_ds at 4 waves per SIMD (128 registers, 5 spilled) and at 3 (170, none) against the separate reduction; its test on the device
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_segmented_bn.py -q -m gpu -k "projection_shortcut" -x 2>&1 | tail -2 | tee gpurun_out/r05x_pytest.txt
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05x_ds_waves.txt; }
for i in 1 2 3; do
  LP_BN_BWD_DS=0 run "separate reduction  "
  run "fused, 4 waves/SIMD "
  LP_HIP_LIB=$PWD/build/liblp_hip_ds3.so run "fused, 3 waves/SIMD "
done
