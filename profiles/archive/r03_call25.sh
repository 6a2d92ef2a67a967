# round 3, call 25: attention without the stored probabilities (lp_attn_fwd_lse / lp_attn_bwd_kv_lse) - device tests, C4 step parity, C4 A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_widen_vitb_width.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r03x_pytest_vit.log; tail -2 gpurun_out/r03x_pytest_vit.log
(timeout 600 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -s -k c4 2>&1 | grep -E "PARITY|passed|failed" | cut -c1-420)
B="timeout 300 python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_ATTN_RECOMPUTE=0 $B 2>&1 | tail -1 | cut -c80-160
  LP_ATTN_RECOMPUTE=1 $B 2>&1 | tail -1 | cut -c80-160
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03x_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r03x_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03x_vit/vit_results.db > gpurun_out/r03x_vit_kernel_stats.txt 2>&1; head -14 gpurun_out/r03x_vit_kernel_stats.txt | cut -c1-60,110-160
