#!/bin/bash
# r05r: conv_pipe_kernel with the next slice's fragment reads interleaved with this slice's MFMAs (-DLP_PIPE_INTERLEAVE=1) against the shipped
# [reads][MFMAs] order, per layer, forward and data gradient, alternating libraries
mkdir -p gpurun_out
for i in 1 2; do
  echo "--- shipped" ; timeout 200 python profiles/conv_layer_bench.py 20
  echo "--- interleaved" ; LP_HIP_LIB=$PWD/build/liblp_hip_il.so timeout 200 python profiles/conv_layer_bench.py 20
done > gpurun_out/r05r_interleave_layers.txt 2>&1
cat gpurun_out/r05r_interleave_layers.txt
