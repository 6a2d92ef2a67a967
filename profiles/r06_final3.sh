# round 6, last check of the final tree (the PMC passes and traces are r06_final2's: conv.hip / conv_pipe.h / conv_res2d.h / conv_stem_wgrad.h / lp_common.h have not
# changed since): whole device suite, smoke, the driver's default bench line, the line without per-launch events, the ViT-S/16 line, serialised kernel traces of both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r06_final3_pytest_gpu.log; tail -2 gpurun_out/r06_final3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160 | tee gpurun_out/r06_final3_smoke.txt
timeout 900 python bench.py > gpurun_out/r06_final3_bench_n1.json.log 2>&1; tail -1 gpurun_out/r06_final3_bench_n1.json.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 20 --warmup 5 > gpurun_out/r06_final3_bench_noprofile.json.log 2>&1; tail -1 gpurun_out/r06_final3_bench_noprofile.json.log | cut -c80-170
timeout 300 python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 4 > gpurun_out/r06_final3_bench_vit.json.log 2>&1; tail -1 gpurun_out/r06_final3_bench_vit.json.log | cut -c80-170
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06_final3_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r06_final3_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06_final3_prof_serial/serial_results.db > gpurun_out/r06_final3_kernel_stats_serial.txt 2>&1; head -8 gpurun_out/r06_final3_kernel_stats_serial.txt | cut -c1-60,110-160
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r06_final3_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r06_final3_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r06_final3_vit/vit_results.db > gpurun_out/r06_final3_vit_kernel_stats.txt 2>&1; head -8 gpurun_out/r06_final3_vit_kernel_stats.txt | cut -c1-60,110-160
