# round 6, closing evidence on the final tree, second take (after the stand-alone HeatmapHead, the GELU fusions and ABI 142; the decode A/B of the
# first take - profiles/r06_final_decode_step_ab.txt - needs its ABI-140 library and is not repeated): whole device suite, smoke, PMC passes (HBM traffic with the kernel-source digest; SQ / MFMA counters),
# the driver's default bench (reads the traffic file written just before), rocprofv3 kernel traces (default + serialised), per-layer table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r06_final2_pytest_gpu.log; tail -2 gpurun_out/r06_final2_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
CMD="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r06_pmc_fetch -o fetch -- $CMD > gpurun_out/r06_pmc_fetch.log 2>&1; tail -1 gpurun_out/r06_pmc_fetch.log | cut -c1-100
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r06_pmc_write -o write -- $CMD > gpurun_out/r06_pmc_write.log 2>&1; tail -1 gpurun_out/r06_pmc_write.log | cut -c1-100
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d /tmp/r06_pmc_sq -o sq -- $CMD > gpurun_out/r06_pmc_sq.log 2>&1; tail -1 gpurun_out/r06_pmc_sq.log | cut -c1-100
python profiles/summarize_pmc.py /tmp/r06_pmc_fetch/fetch_results.db /tmp/r06_pmc_write/write_results.db > gpurun_out/r06_pmc_traffic.json 2> gpurun_out/r06_pmc_traffic.err; head -4 gpurun_out/r06_pmc_traffic.json
python profiles/summarize_pmc_sq.py /tmp/r06_pmc_sq/sq_results.db > gpurun_out/r06_pmc_mfma.json 2> gpurun_out/r06_pmc_mfma.err; tail -2 gpurun_out/r06_pmc_mfma.err
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
timeout 900 python bench.py > gpurun_out/r06_final2_bench_n1.json.log 2>&1; tail -1 gpurun_out/r06_final2_bench_n1.json.log | cut -c1-330
# (the per-layer table from a headline-only run: the default run's secondary configurations overwrite the launch dump)
LP_DUMP_LAUNCHES=gpurun_out/r06_final2_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r06_final2_bench_layers.json.log 2>&1
python profiles/layer_table.py gpurun_out/r06_final2_launches.json > gpurun_out/r06_final2_layer_table.txt 2>&1; tail -1 gpurun_out/r06_final2_layer_table.txt
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r06_final2_bench_noprofile.json.log 2>&1; tail -1 gpurun_out/r06_final2_bench_noprofile.json.log | cut -c80-160
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06_final2_prof -o dflt -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r06_final2_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06_final2_prof/dflt_results.db > gpurun_out/r06_final2_kernel_stats.txt 2>&1
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06_final2_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r06_final2_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06_final2_prof_serial/serial_results.db > gpurun_out/r06_final2_kernel_stats_serial.txt 2>&1; head -12 gpurun_out/r06_final2_kernel_stats_serial.txt | cut -c1-60,110-160
python profiles/gap_analysis.py /tmp/r06_final2_prof/dflt_results.db > gpurun_out/r06_final2_gap_analysis.txt 2>&1; head -9 gpurun_out/r06_final2_gap_analysis.txt
python profiles/stream_tail.py /tmp/r06_final2_prof/dflt_results.db > gpurun_out/r06_final2_stream_tail.txt 2>&1; head -4 gpurun_out/r06_final2_stream_tail.txt
timeout 300 python profiles/decode_microbench.py > gpurun_out/r06_final2_decode.txt 2>/dev/null; tail -4 gpurun_out/r06_final2_decode.txt | cut -c1-140
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06_final2_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r06_final2_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06_final2_vit/vit_results.db > gpurun_out/r06_final2_vit_kernel_stats.txt 2>&1; head -3 gpurun_out/r06_final2_vit_kernel_stats.txt | cut -c1-60,110-160
