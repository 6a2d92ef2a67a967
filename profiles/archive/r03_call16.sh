# round 3, call 16: the bf16-mixed POLICY's own gradients at BASELINE's real batch (reference arithmetic rounded where the product rounds, torch on the device:
# the autograd state of 192 frames does not fit the build container) - the yardstick for the product's stem-gradient cosine in test_step_parity[c2full-bf16-mixed]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DEVICE=cuda:0 timeout 900 python profiles/policy_grad_full.py c2full > gpurun_out/r03_policy_grad_c2full.json 2> gpurun_out/r03_policy_grad_c2full.err; tail -3 gpurun_out/r03_policy_grad_c2full.err; head -c 1500 gpurun_out/r03_policy_grad_c2full.json
