#!/bin/bash
# r06a (round 6, first call): the whole device suite on the round's tree (advisor fixes, trajectory test), then the decode kernels with the row tables
# read by SCALAR loads (lp_common.h: as_uniform) against the per-lane loads of rounds 1 - 5 (build/liblp_hip_nouni.so, -DLP_UNIFORM_LOADS=0):
# microbench + step A/B (alternating processes), the torch-glue trace of one step, kernel stats of the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider -x 2>&1 | tail -40) > gpurun_out/r06a_pytest_gpu.log; tail -3 gpurun_out/r06a_pytest_gpu.log
(timeout 600 python -m pytest tests/test_trajectory_vs_reference.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -v "^    " | tail -80) > gpurun_out/r06a_trajectory.txt; tail -3 gpurun_out/r06a_trajectory.txt
for lib in nouni new; do
  echo "== decode microbench, library: $lib" | tee -a gpurun_out/r06a_decode_ab.txt
  if [ $lib = nouni ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nouni.so; else unset LP_HIP_LIB; fi
  timeout 300 python profiles/decode_microbench.py 2>&1 | tee -a gpurun_out/r06a_decode_ab.txt
done
for i in 1 2 3; do
  for lib in nouni new; do
    if [ $lib = nouni ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nouni.so; else unset LP_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06a_step_ab.txt
  done
done
unset LP_HIP_LIB
timeout 300 python profiles/glue_trace.py > gpurun_out/r06a_glue_trace.txt 2>&1; tail -30 gpurun_out/r06a_glue_trace.txt
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06a_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r06a_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06a_prof_serial/serial_results.db > gpurun_out/r06a_kernel_stats_serial.txt 2>&1; head -14 gpurun_out/r06a_kernel_stats_serial.txt | cut -c1-60,110-160
