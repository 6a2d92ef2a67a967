# round 3, call 10: timing experiments on conv_pipe_kernel's K loop (variant builds that skip the loads / the per-slice LDS reads / the barriers:
# wrong results, the time is the point)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S="l2.c2:192:48:128:128:3:1:1 l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1 l4.c2:192:12:512:512:3:1:1"
for v in base exp_NOLOAD exp_NOLDSREAD exp_NOBARRIER exp_NOLOAD_NOBARRIER; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $v"; LP_HIP_LIB=$lib KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd
done > gpurun_out/r03j_loop_experiments.txt 2>&1
cat gpurun_out/r03j_loop_experiments.txt
