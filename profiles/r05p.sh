#!/bin/bash
# r05p: the stem's weight gradient straight from the pooled gradient (lp_stem_wgrad_pool; LP_STEM_WGRAD_POOL=0 = lp_bn_pool_bwd_apply + lp_stem_wgrad):
# its tests on the device, the step A/B (alternating processes), the tail of the step's timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_conv.py tests/test_emu_engine.py -q -m gpu -k "stem or pooled or blockwise" -x 2>&1 | tail -3 | tee gpurun_out/r05p_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_STEM_WGRAD_POOL=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('POOL=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05p_step_ab.txt
  done
done
timeout 300 rocprofv3 --kernel-trace -d /tmp/r05p_prof -o t -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r05p_prof.log 2>&1
python profiles/stream_tail.py /tmp/r05p_prof/t_results.db > gpurun_out/r05p_stream_tail.txt 2>&1
tail -24 gpurun_out/r05p_stream_tail.txt
