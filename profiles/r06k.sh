#!/bin/bash
# r06k: both GELU fusions (lp_gemm_nt_gelu_fwd / _bwd) with the erfc-fit GELU (lp_common.h: gelu_phi) on the device: tests incl. the C4 step fixtures,
# stand-alone HeatmapHead with every option, the tightened PCA-fit bars; then the ViT-S/16 step (BASELINE C4) with and without the fusion,
# alternating processes, and a kernel trace of the fused step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_head_options.py tests/test_differential_vs_reference.py tests/test_step_parity.py tests/test_widen_vitb_width.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4) | tee gpurun_out/r06k_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_VIT_GELU_FUSED=$m timeout 300 python bench.py --backbone vits_dino --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VIT_GELU_FUSED=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06k_vit_step_ab.txt
  done
done
rm -rf /tmp/r06k_prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r06k_prof -o vit -- python bench.py --backbone vits_dino --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
python profiles/summarize_rocpd.py $(ls /tmp/r06k_prof/*results.db /tmp/r06k_prof/*/*results.db 2>/dev/null | head -1) > gpurun_out/r06k_vit_kernel_stats.txt 2>&1
head -14 gpurun_out/r06k_vit_kernel_stats.txt | cut -c1-150
