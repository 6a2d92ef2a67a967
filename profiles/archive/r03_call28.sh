# round 3, call 28: the ViT's bias gradients taken by the kernels that produce dy (gelu / LayerNorm backward), weight gradients without the bias pass
# (fc1 / fc2 / proj on the pipelined kernel): device tests, C4 step parity, C4 A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_widen_vitb_width.py tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -k "vit or c4" 2>&1 | tail -3) > gpurun_out/r03aa_pytest.log; tail -2 gpurun_out/r03aa_pytest.log
B="timeout 300 python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2 3; do
  for f in 0 1; do echo -n "bias_fused=$f "; LP_VIT_BIAS_FUSED=$f $B 2>&1 | tail -1 | cut -c88-110; done
done > gpurun_out/r03aa_vit_bias.txt 2>&1; cat gpurun_out/r03aa_vit_bias.txt
