#!/bin/bash
# r06m: the GELU fusions with the packed-pair arithmetic (v_pk_fma_f32; A&S form in the backward: one exponential): none / backward only / both, alternating processes; ViT tests; trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  for m in 0 bwd 1; do
    LP_VIT_GELU_FUSED=$m timeout 300 python bench.py --backbone vits_dino --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VIT_GELU_FUSED=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06m_vit_step_ab.txt
  done
done
(timeout 1800 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_step_parity.py tests/test_widen_vitb_width.py -q -m gpu -x -p no:cacheprovider -k "vit or c4 or gelu or gemm" 2>&1 | tail -3) | tee gpurun_out/r06m_pytest.txt
rm -rf /tmp/r06m_prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r06m_prof -o vit -- python bench.py --backbone vits_dino --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
python profiles/summarize_rocpd.py $(ls /tmp/r06m_prof/*results.db /tmp/r06m_prof/*/*results.db 2>/dev/null | head -1) > gpurun_out/r06m_vit_kernel_stats.txt 2>&1
head -16 gpurun_out/r06m_vit_kernel_stats.txt | cut -c1-150
