#!/bin/bash
# r06w: softmax2d_bwd_pixmajor_kernel with its result rows staged per wave in LDS and stored as coalesced 1-KB pieces, against lane-owned rows (16 B per lane at a
# 128-B stride; build/liblp_hip_oldsm.so = this tree's objects with the previous heatmap.hip): kernel durations from serialised traces, the step alternating, tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_losses.py tests/test_emu_engine.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2) | tee gpurun_out/r06w_pytest.txt
for lib in oldsm new; do
  if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
  rm -rf /tmp/r06w_prof
  LP_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r06w_prof -o t -- python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
  python profiles/summarize_rocpd.py $(ls /tmp/r06w_prof/*results.db /tmp/r06w_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "softmax2d" | cut -c1-70,100-170 | sed "s/^/$lib /" | tee -a gpurun_out/r06w_softmax_kernels.txt
done
for i in 1 2 3; do
  for lib in oldsm new; do
    if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('softmax_bwd=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06w_step_ab.txt
  done
done
