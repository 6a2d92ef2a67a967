# round 3, call 17: stores of the previous tile no longer waited for at tile boundaries (counted vmcnt relaxed by the stores issued after the needed loads):
# parity on real shapes (3 repeats each), per-layer A/B against the strict build, step A/B; the policy's own gradients at the full batch (retry)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_emu_conv_pipe.py tests/test_emu_conv.py -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -5) > gpurun_out/r03p_pytest.log; tail -3 gpurun_out/r03p_pytest.log
S="l1.c1:192:96:256:64:1:1:0 l1.c2:192:96:64:64:3:1:1 l1.c3:192:96:64:256:1:1:0 l2.c1:192:48:512:128:1:1:0 l2.c2:192:48:128:128:3:1:1 l2.c3:192:48:128:512:1:1:0 l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1 l3.c3:192:24:256:1024:1:1:0 l4.c2:192:12:512:512:3:1:1"
for rep in 1 2; do for v in strictvm base; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $v"; LP_HIP_LIB=$lib timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep -E "fwd|dgrad"
done; done > gpurun_out/r03p_vmcnt_layers.txt 2>&1; cat gpurun_out/r03p_vmcnt_layers.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_strictvm.so $B 2>&1 | tail -1 | cut -c80-160
  $B 2>&1 | tail -1 | cut -c80-160
done
DEVICE=cuda:0 timeout 900 python profiles/policy_grad_full.py c2full > gpurun_out/r03_policy_grad_c2full.json 2> gpurun_out/r03_policy_grad_c2full.err; tail -3 gpurun_out/r03_policy_grad_c2full.err; head -c 1200 gpurun_out/r03_policy_grad_c2full.json
