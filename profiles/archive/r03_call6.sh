# round 3, call 6: why conv_wgrad_pipe_kernel's K steps are slow - PMC over single-layer launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KINDS=wgrad timeout 120 python profiles/conv_layer_bench.py 5 > gpurun_out/r03f_layer_bench_wgpipe.txt 2>&1; cat gpurun_out/r03f_layer_bench_wgpipe.txt
KINDS=wgrad LP_WGRAD_PIPE=0 timeout 120 python profiles/conv_layer_bench.py 5 > gpurun_out/r03f_layer_bench_wgold.txt 2>&1; cat gpurun_out/r03f_layer_bench_wgold.txt
for spec in l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1; do
  n=${spec%%:*}
  KINDS=wgrad timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/pmcw_$n -o sq -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03f_pmc_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmcw_$n/sq_results.db wgrad_pipe > gpurun_out/r03f_pmc_sq_$n.json 2>&1
  KINDS=wgrad timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d /tmp/pmcw2_$n -o sq -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03f_pmc2_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmcw2_$n/sq_results.db wgrad_pipe > gpurun_out/r03f_pmc_sq2_$n.json 2>&1
  KINDS=wgrad timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d /tmp/pmcw3_$n -o tcc -- python profiles/conv_layer_bench.py 3 $spec > gpurun_out/r03f_pmc3_$n.log 2>&1
  python profiles/summarize_pmc_any.py /tmp/pmcw3_$n/tcc_results.db wgrad_pipe > gpurun_out/r03f_pmc_tcc_$n.json 2>&1
  echo "== $n"; cat gpurun_out/r03f_pmc_sq_$n.json gpurun_out/r03f_pmc_sq2_$n.json gpurun_out/r03f_pmc_tcc_$n.json | tr -d '\n ' | cut -c1-1500; echo
done
