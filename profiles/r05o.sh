#!/bin/bash
# r05o: a layer's weight gradient enqueued behind its data gradient (LP_WGRAD_AFTER_DGRAD=1: an engine switch of commit 6b4136e..b1d58a4, removed after this measurement) against the default order, alternating processes
mkdir -p gpurun_out
for i in 1 2 3; do
  for m in 0 1; do
    LP_WGRAD_AFTER_DGRAD=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AFTER_DGRAD=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05o_wgrad_order_ab.txt
  done
done
