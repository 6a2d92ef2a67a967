# round 3, call 20: which small launches does a step enqueue besides the lp_* kernels (torch.profiler, Python stacks); weight gradients of the ViT's Linear
# layers on the pipelined kernel? (per-layer A/B without the bias gradient); cubic resize + parity re-check on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python profiles/glue_trace.py > gpurun_out/r03s_glue_trace.txt 2>&1; head -70 gpurun_out/r03s_glue_trace.txt | cut -c1-170
S="qkv:192:24:384:1152:1:1:0 proj:192:24:384:384:1:1:0 fc1:192:24:384:1536:1:1:0 fc2:192:24:1536:384:1:1:0"
for w in 0 1 2; do echo "== LP_WGRAD_PIPE=$w"; LP_WGRAD_PIPE=$w KINDS=wgrad,fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep -E "wgrad|fwd"; done > gpurun_out/r03s_vit_wgrad.txt 2>&1; cat gpurun_out/r03s_vit_wgrad.txt
(timeout 600 python -m pytest tests/test_widen_n1n2_kernels.py tests/test_widen_n1n2_stack.py tests/test_widen_n2_dataset.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r03s_pytest_n2.log; tail -2 gpurun_out/r03s_pytest_n2.log
