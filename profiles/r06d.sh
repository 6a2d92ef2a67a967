#!/bin/bash
# r06d: the decode's moments about the tile maximum (default) against moments from the corner (build/liblp_hip_nocenter.so, -DLP_DEC_CENTER=0): the forward
# kernel read 402 -> 463 us between two calls (two boxes) - here both libraries alternate inside one call; trajectory test with its final bars;
# per-layer table with layer names; PMC of decode_fwd for both builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for lib in nocenter new; do
    if [ $lib = nocenter ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nocenter.so; else unset LP_HIP_LIB; fi
    echo "== decode microbench, library: $lib" | tee -a gpurun_out/r06d_decode_ab.txt
    timeout 300 python profiles/decode_microbench.py 2>/dev/null | tee -a gpurun_out/r06d_decode_ab.txt
  done
done
for i in 1 2 3; do
  for lib in nocenter new; do
    if [ $lib = nocenter ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nocenter.so; else unset LP_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06d_step_ab.txt
  done
done
unset LP_HIP_LIB
(timeout 600 python -m pytest tests/test_trajectory_vs_reference.py tests/test_boundary_reference_factory.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep "step \|passed\|failed" | head -60) > gpurun_out/r06d_trajectory.txt; tail -2 gpurun_out/r06d_trajectory.txt
LP_DUMP_LAUNCHES=gpurun_out/r06d_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r06d_bench_layers.json.log 2>&1
python profiles/layer_table.py gpurun_out/r06d_launches.json > gpurun_out/r06d_layer_table.txt 2>&1; tail -3 gpurun_out/r06d_layer_table.txt
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD"
for lib in nocenter new; do
  if [ $lib = nocenter ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nocenter.so; else unset LP_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $P2 -d /tmp/r06d_pmc_$lib -o p -- python profiles/decode_microbench.py > gpurun_out/r06d_pmc_$lib.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/r06d_pmc_$lib/p_results.db decode_fwd > gpurun_out/r06d_decode_fwd_pmc_$lib.json 2>> gpurun_out/r06d_pmc_$lib.log
done
unset LP_HIP_LIB
