#!/bin/bash
# r05s: interleaved fragment reads (r05r) with two or three fragment sets, the 3x3 / deep 1x1 layers, forward; then the step with the interleaved build
mkdir -p gpurun_out
L="l2.c2:192:48:128:128:3:1:1 l3.c2:192:24:256:256:3:1:1 l4.c2:192:12:512:512:3:1:1 l3.c1:192:24:1024:256:1:1:0 l4.c3:192:12:512:2048:1:1:0"
for i in 1 2; do
  echo "--- shipped" ; KINDS=fwd timeout 200 python profiles/conv_layer_bench.py 30 $L
  echo "--- interleaved, 2 sets" ; KINDS=fwd LP_HIP_LIB=$PWD/build/liblp_hip_il.so timeout 200 python profiles/conv_layer_bench.py 30 $L
  echo "--- interleaved, 3 sets" ; KINDS=fwd LP_HIP_LIB=$PWD/build/liblp_hip_il3.so timeout 200 python profiles/conv_layer_bench.py 30 $L
done > gpurun_out/r05s_interleave_sets.txt 2>&1
cat gpurun_out/r05s_interleave_sets.txt
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05s_step_ab.txt; }
for i in 1 2 3; do
  run "shipped     "
  LP_HIP_LIB=$PWD/build/liblp_hip_il.so run "interleaved "
done
