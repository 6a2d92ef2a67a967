#!/bin/bash
# r05k: cycle-counter probe of the forward convolution kernels (timing build in build/, see profiles/conv_timing_probe.py)
set -x
mkdir -p gpurun_out
export LP_HIP_LIB=$PWD/build/liblp_hip_timing.so
KINDS=fwd,dgrad timeout 600 python profiles/conv_timing_probe.py > gpurun_out/r05k_conv_timing.txt 2>&1
tail -80 gpurun_out/r05k_conv_timing.txt
