#!/bin/bash
# r05aa: the spatial soft-max of the head, forward and backward, one workgroup per (frame, group of 8 maps) instead of one per frame (192 workgroups on 256
# CUs): step A/B against the previous build (build/liblp_hip_smold.so), alternating processes; kernel times of both from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_losses.py -q -m gpu -x 2>&1 | tail -2 | tee gpurun_out/r05aa_pytest.txt
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05aa_step_ab.txt; }
for i in 1 2 3; do
  LP_HIP_LIB=$PWD/build/liblp_hip_smold.so run "one workgroup per frame   "
  run "per (frame, 8 maps)       "
done
for v in old new; do
  L=""; [ $v = old ] && L=$PWD/build/liblp_hip_smold.so
  LP_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r05aa_$v -o t -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > /dev/null 2>&1
  python profiles/summarize_rocpd.py /tmp/r05aa_$v/t_results.db 2>/dev/null | grep -i "softmax2d" | cut -c1-60,110-175 | sed "s/^/$v: /" | tee -a gpurun_out/r05aa_kernel_times.txt
done
