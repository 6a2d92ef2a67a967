#!/bin/bash
# r06p: the stem's BatchNorm + ReLU + max-pool forward with its nine taps requested before the first is used (bn.hip, round 6) against the kernel whose
# border branches made every tap a dependent load (build/liblp_hip_oldpool.so = this tree's objects with the previous bn.hip linked in): the kernel
# alone at 64 and 128 frames, the step (alternating processes), the kernel tests on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_trunk_ops.py tests/test_emu_ragged_shapes.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3) | tee gpurun_out/r06p_pytest.txt
for lib in oldpool new; do
  if [ $lib = oldpool ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_oldpool.so; else unset LP_HIP_LIB; fi
  for b in 64 128; do timeout 200 python profiles/pool_microbench.py $b 2>/dev/null | sed "s/^/$lib B=$b /" | tee -a gpurun_out/r06p_pool_microbench.txt; done
done
for i in 1 2 3; do
  for lib in oldpool new; do
    if [ $lib = oldpool ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_oldpool.so; else unset LP_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool_fwd=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06p_step_ab.txt
  done
done
