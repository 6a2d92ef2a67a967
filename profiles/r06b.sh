#!/bin/bash
# r06b: (1) lp_bn_finalize folded into the apply launch (lp_bn_apply_seg_fin; LP_BN_FIN_FUSED=0 = the stand-alone launch): device tests + step A/B;
# (2) where decode_bwd_kernel's time goes: timing builds with parts of the kernel compiled out (-DLP_DEC_PROBE=1 no strip product, 2 no per-row
# arithmetic, 6 = 2 + no window values, 8 no phase A at all) and SQ counters of the microbench's launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_step_parity.py tests/test_emu_decode.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4) | tee gpurun_out/r06b_pytest.txt
for i in 1 2 3; do
  for m in 0 1; do
    LP_BN_FIN_FUSED=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_FIN_FUSED=$m', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06b_step_ab.txt
  done
done
for v in full dp1 dp2 dp6 dp8; do
  if [ $v = full ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; fi
  echo "== decode microbench, build: $v" | tee -a gpurun_out/r06b_decode_probe.txt
  timeout 300 python profiles/decode_microbench.py 2>/dev/null | grep '"prune": 0' | tee -a gpurun_out/r06b_decode_probe.txt
done
unset LP_HIP_LIB
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_IFETCH"
n=1
for P in "$P1" "$P2" "$P3"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d /tmp/r06b_pmc$n -o p -- python profiles/decode_microbench.py > gpurun_out/r06b_pmc$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/r06b_pmc$n/p_results.db decode > gpurun_out/r06b_decode_pmc$n.json 2>> gpurun_out/r06b_pmc$n.log
  n=$((n+1))
done
head -c 1500 gpurun_out/r06b_decode_pmc1.json
