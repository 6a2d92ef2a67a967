# round 3, call 4: what the deep layers' K steps wait for - PMC passes over single-layer launches of conv_pipe_kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python profiles/conv_layer_bench.py 5 > gpurun_out/r03d_layer_bench.txt 2>&1; cat gpurun_out/r03d_layer_bench.txt
for spec in l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1 l4.c3:192:12:512:2048:1:1:0 l1.c3:192:96:64:256:1:1:0; do
  n=${spec%%:*}
  KINDS=fwd timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/pmc_$n -o sq -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03d_pmc_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmc_$n/sq_results.db conv_pipe > gpurun_out/r03d_pmc_sq_$n.json 2>&1
  KINDS=fwd timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d /tmp/pmc2_$n -o sq -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03d_pmc2_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmc2_$n/sq_results.db conv_pipe > gpurun_out/r03d_pmc_sq2_$n.json 2>&1
  KINDS=fwd timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d /tmp/pmc3_$n -o tcc -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03d_pmc3_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmc3_$n/tcc_results.db conv_pipe > gpurun_out/r03d_pmc_tcc_$n.json 2>&1
  echo "== $n"; cat gpurun_out/r03d_pmc_sq_$n.json gpurun_out/r03d_pmc_sq2_$n.json gpurun_out/r03d_pmc_tcc_$n.json | tr -d '\n ' | cut -c1-1500; echo
done
