# round 3, call 11: the two waves of a SIMD do their per-K-step address arithmetic at different points (forward kernel) - A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S="l1.c2:192:96:64:64:3:1:1 l2.c2:192:48:128:128:3:1:1 l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1 l3.c3:192:24:256:1024:1:1:0 l4.c2:192:12:512:512:3:1:1 l4.c3:192:12:512:2048:1:1:0"
for rep in 1 2; do for v in base nostagger; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $v"; LP_HIP_LIB=$lib KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd
done; done > gpurun_out/r03k_stagger.txt 2>&1
cat gpurun_out/r03k_stagger.txt
(timeout 300 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py -q -m gpu --timeout 300 -p no:cacheprovider -x -k "pipe or pipelined" 2>&1 | tail -3)
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  $B 2>&1 | tail -1 | cut -c80-160
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nostagger.so $B 2>&1 | tail -1 | cut -c80-160
done
