# round 3, call 35: ViT bias fusion refined (proj back inside its weight-gradient launch, fp32 column sums): C4 A/B, parity, trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -k "vit or c4" 2>&1 | tail -2)
B="timeout 300 python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2 3; do
  for f in 0 1; do echo -n "bias_fused=$f "; LP_VIT_BIAS_FUSED=$f $B 2>&1 | tail -1 | cut -c88-110; done
done > gpurun_out/r03af_vit_bias2.txt 2>&1; cat gpurun_out/r03af_vit_bias2.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03_final_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r03_final_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03_final_vit/vit_results.db > gpurun_out/r03_final_vit_kernel_stats.txt 2>&1; head -12 gpurun_out/r03_final_vit_kernel_stats.txt | cut -c1-60,110-160
