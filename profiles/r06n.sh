#!/bin/bash
# r06n: LayerNorm with half a wave per row (vit.hip, round 6) against the one-wave-per-row kernels (build/liblp_hip_oldln.so = this tree's objects with
# the previous vit.hip linked in), alternating processes, ViT-S/16 step (BASELINE C4); the ViT tests on the device; a kernel trace of the new step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_step_parity.py tests/test_widen_vitb_width.py -q -m gpu -x -p no:cacheprovider -k "vit or c4 or layernorm" 2>&1 | tail -3) | tee gpurun_out/r06n_pytest.txt
for i in 1 2 3; do
  for lib in oldln new; do
    if [ $lib = oldln ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_oldln.so; else unset LP_HIP_LIB; fi
    timeout 300 python bench.py --backbone vits_dino --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('layernorm=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06n_vit_step_ab.txt
  done
done
unset LP_HIP_LIB
rm -rf /tmp/r06n_prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r06n_prof -o vit -- python bench.py --backbone vits_dino --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
python profiles/summarize_rocpd.py $(ls /tmp/r06n_prof/*results.db /tmp/r06n_prof/*/*results.db 2>/dev/null | head -1) > gpurun_out/r06n_vit_kernel_stats.txt 2>&1
grep -i "layernorm\|gelu" gpurun_out/r06n_vit_kernel_stats.txt | cut -c1-60,100-160
