#!/bin/bash
# r06z: attn_fwd_kernel with its soft-max in base 2 (one v_exp_f32 per exponential, the scale folded into an FMA) against exp(x) = multiply + v_exp
# (build/liblp_hip_oldattn.so = this tree with the previous attn.hip): ViT tests on the device, kernel durations from traces, the ViT-S/16 step alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_emu_vit_ops.py tests/test_emu_vit_engine.py tests/test_step_parity.py tests/test_widen_vitb_width.py tests/test_widen_inference_stack.py tests/test_widen_inference_kernels.py -q -m gpu -x -p no:cacheprovider -k "vit or c4 or attn or attention" 2>&1 | tail -2) | tee gpurun_out/r06z_pytest.txt
for lib in oldattn new; do
  if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
  rm -rf /tmp/r06z_prof
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r06z_prof -o t -- python bench.py --backbone vits_dino --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
  python profiles/summarize_rocpd.py $(ls /tmp/r06z_prof/*results.db /tmp/r06z_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "attn_" | cut -c1-50,100-170 | sed "s/^/$lib /" | tee -a gpurun_out/r06z_attn_kernels.txt
done
for i in 1 2 3; do
  for lib in oldattn new; do
    if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
    timeout 300 python bench.py --backbone vits_dino --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_rows=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06z_vit_step_ab.txt
  done
done
