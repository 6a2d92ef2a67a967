# round 3, call 18: what bounds the K loop of the HALO form (64-channel layer1 conv2 at 250 us against a 52 us MFMA floor)? loop experiments (wrong results:
# no loads / fragments read once per K step / no barriers) and 3 fragment sets, forward, HALO on and off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S="l1.c2:192:96:64:64:3:1:1 l2.c2:192:48:128:128:3:1:1 l3.c2:192:24:256:256:3:1:1"
for v in base exp_noload exp_noldsread exp_nobarrier frag3; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  for h in 1 0; do echo "== $v halo=$h"; LP_CONV_HALO=$h LP_HIP_LIB=$lib KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd; done
done > gpurun_out/r03q_halo_loop_experiments.txt 2>&1; cat gpurun_out/r03q_halo_loop_experiments.txt
