#!/bin/bash
# r05z: the projection shortcut's data gradient on its own grid, conv1's data gradient the last writer (adds it at the even pixels, takes the ReLU mask and
# the previous block's BatchNorm reductions: lp_bn_fuse.addend_half) against conv1 first + in-place accumulation + stand-alone reduction (LP_DGRAD_HALF_ADDEND=0)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_conv_pipe.py tests/test_emu_engine.py tests/test_step_parity.py -q -m gpu -k "shortcut_gradient or blockwise or c2full or c1 or c5" -x 2>&1 | tail -3 | tee gpurun_out/r05z_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_DGRAD_HALF_ADDEND=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HALF_ADDEND=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05z_step_ab.txt
  done
done
