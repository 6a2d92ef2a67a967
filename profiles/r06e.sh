#!/bin/bash
# r06e: the tree after the decode rework (no LDS atomics, moments about the tile maximum, padded transposed store) and the retired BatchNorm-finalisation fusion:
# whole device suite, trajectory + boundary tests with their final bars, decode microbench, three plain step timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r06e_pytest_gpu.log; tail -3 gpurun_out/r06e_pytest_gpu.log
(timeout 900 python -m pytest tests/test_trajectory_vs_reference.py tests/test_boundary_reference_factory.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep "^  step \|^trajectory\|passed\|failed" | cut -c1-170) > gpurun_out/r06e_trajectory.txt; tail -2 gpurun_out/r06e_trajectory.txt
timeout 300 python profiles/decode_microbench.py 2>/dev/null | tee gpurun_out/r06e_decode.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06e_step.txt
done
