# round 2: PMC passes of the default bench command (each counter group in its own run, with --kernel-trace only):
#   A FETCH_SIZE   B WRITE_SIZE   C SQ MFMA / issue counters   -> profiles/r02_pmc_traffic.json, profiles/r02_pmc_mfma.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1"
rocprofv3 -L 2>/dev/null | grep -E "SQ_VALU_MFMA|SQ_INSTS_VALU_MFMA|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_ANY|SQ_LDS_BANK|SQ_WAIT_ANY|SQ_BUSY_CU|GRBM_GUI_ACTIVE|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_INST_CYCLES_VMEM" | sort -u | head -40 > gpurun_out/r02_pmc_available.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r02_pmc_fetch -o fetch -- $CMD > gpurun_out/r02_pmc_fetch.log 2>&1; tail -1 gpurun_out/r02_pmc_fetch.log | cut -c1-120
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r02_pmc_write -o write -- $CMD > gpurun_out/r02_pmc_write.log 2>&1; tail -1 gpurun_out/r02_pmc_write.log | cut -c1-120
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d /tmp/r02_pmc_sq -o sq -- $CMD > gpurun_out/r02_pmc_sq.log 2>&1; tail -1 gpurun_out/r02_pmc_sq.log | cut -c1-120
python profiles/summarize_pmc.py /tmp/r02_pmc_fetch/fetch_results.db /tmp/r02_pmc_write/write_results.db > gpurun_out/r02_pmc_traffic.json 2> gpurun_out/r02_pmc_traffic.err; head -5 gpurun_out/r02_pmc_traffic.json
python profiles/summarize_pmc_sq.py /tmp/r02_pmc_sq/sq_results.db > gpurun_out/r02_pmc_mfma.json 2> gpurun_out/r02_pmc_mfma.err; head -30 gpurun_out/r02_pmc_mfma.json; cat gpurun_out/r02_pmc_mfma.err | tail -3
