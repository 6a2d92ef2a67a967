#!/bin/bash
# r05v: does the BatchNorm backward overlap the side stream's weight gradients better with fewer registers (more of its waves fit beside a weight-gradient
# workgroup: 86 registers -> 2 waves per SIMD beside conv_wgrad_pipe_kernel, 0 beside conv_wgrad_kernel<128>; 80 / 72 / 64 -> 2 / 3 / 3 and 1 / 1 / 1)?
# -DLP_BN_BWD_WAVES=6 / 7 / 8 builds (7 / 16 / 27 registers spilled) against the shipped kernel; alternating processes
mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05v_bn_bwd_regs.txt; }
for i in 1 2; do
  run "shipped (86 regs)"
  for w in 6 7 8; do LP_HIP_LIB=$PWD/build/liblp_hip_bw$w.so run "waves/SIMD $w        "; done
done
