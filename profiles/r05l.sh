#!/bin/bash
# r05l: stem weight gradient from the staged neighbourhood - kernel A/B, its tests on the device, then the step A/B (alternating processes)
set -x
mkdir -p gpurun_out
timeout 300 python profiles/stem_wgrad_bench.py > gpurun_out/r05l_stem_wgrad.txt 2>&1
cat gpurun_out/r05l_stem_wgrad.txt
timeout 600 python -m pytest tests/test_emu_conv.py -q -m gpu -k stem -x 2>&1 | tail -3 | tee gpurun_out/r05l_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_STEM_WGRAD_NB=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NB=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05l_step_ab.txt
  done
done
