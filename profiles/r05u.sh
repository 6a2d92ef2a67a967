#!/bin/bash
# r05u: the two data gradients into a layer's first block read the block input's ReLU mask at 1 bit per element (lp_conv_dgrad_bits, kEkPB) instead of
# the bf16 activation (LP_DGRAD_MASK_BITS=0): tests on the device, step A/B (alternating processes), per-layer table of both
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_conv_pipe.py tests/test_emu_engine.py tests/test_step_parity.py -q -m gpu -k "pipe_equals_igemm or blockwise or c2full or c1" -x 2>&1 | tail -3 | tee gpurun_out/r05u_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_DGRAD_MASK_BITS=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MASK_BITS=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05u_step_ab.txt
  done
done
for m in 0 1; do
  LP_DGRAD_MASK_BITS=$m LP_DUMP_LAUNCHES=gpurun_out/r05u_launches_$m.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > /dev/null 2>&1
  python profiles/layer_table.py gpurun_out/r05u_launches_$m.json > gpurun_out/r05u_layer_table_bits$m.txt 2>&1
  grep -E "l[234]\.0\.(c1|down)|totals" gpurun_out/r05u_layer_table_bits$m.txt
done
