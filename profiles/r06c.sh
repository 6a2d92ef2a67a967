#!/bin/bash
# r06c: decode_bwd_kernel without LDS atomics (owner stores / guest adds) + moments about the tile's maximum; lp_bn_apply_seg_fin inside its register cap:
# device tests, decode microbench (r06b's 1448 us / 748 us pruned are the "before"), step A/B of the fused finalisation, kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r06c_pytest_gpu.log; tail -3 gpurun_out/r06c_pytest_gpu.log
(timeout 600 python -m pytest tests/test_trajectory_vs_reference.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep "step \|passed\|failed" | head -60) > gpurun_out/r06c_trajectory.txt; tail -2 gpurun_out/r06c_trajectory.txt
timeout 300 python profiles/decode_microbench.py 2>/dev/null | tee gpurun_out/r06c_decode.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_BN_FIN_FUSED=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_FIN_FUSED=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06c_step_ab.txt
  done
done
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06c_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r06c_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06c_prof_serial/serial_results.db > gpurun_out/r06c_kernel_stats_serial.txt 2>&1; grep -i "decode\|finalize\|bn_apply" gpurun_out/r06c_kernel_stats_serial.txt | cut -c1-60,110-170
