#!/bin/bash
# r06o: the head's soft-max kernels with four workgroups per frame (heatmap.hip, round 6) against one per frame (build/liblp_hip_oldsm.so = this tree's
# objects with the previous heatmap.hip linked in): the headline step, alternating processes; kernel tests on the device; kernel durations from a trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_losses.py tests/test_emu_engine.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3) | tee gpurun_out/r06o_pytest.txt
for i in 1 2 3; do
  for lib in oldsm new; do
    if [ $lib = oldsm ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_oldsm.so; else unset LP_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('softmax=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06o_step_ab.txt
  done
done
for lib in oldsm new; do
  if [ $lib = oldsm ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_oldsm.so; else unset LP_HIP_LIB; fi
  rm -rf /tmp/r06o_prof
  LP_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r06o_prof -o t -- python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
  python profiles/summarize_rocpd.py $(ls /tmp/r06o_prof/*results.db /tmp/r06o_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "softmax2d" | cut -c1-70,100-170 | sed "s/^/$lib /" | tee -a gpurun_out/r06o_softmax_kernels.txt
done
