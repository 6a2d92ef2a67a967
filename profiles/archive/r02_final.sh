# round 2, final evidence run on the final tree: the device suite (twice: flake check), smoke, the driver's default bench command, its rocprofv3
# kernel trace (default = weight gradients on the side stream, and serialised), ViT trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r02_final_pytest_gpu.log; tail -2 gpurun_out/r02_final_pytest_gpu.log
(timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r02_final_pytest_gpu_again.log; tail -1 gpurun_out/r02_final_pytest_gpu_again.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2) > gpurun_out/r02_final_smoke.log; tail -1 gpurun_out/r02_final_smoke.log
timeout 600 python bench.py > gpurun_out/r02_final_bench_n1.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_n1.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02_final_bench_n1_noprofile.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_n1_noprofile.json.log | cut -c1-200
LP_DUMP_LAUNCHES=gpurun_out/r02_final_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > /dev/null 2>&1
python profiles/layer_table.py gpurun_out/r02_final_launches.json > gpurun_out/r02_final_layer_table.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r02_final_prof -o final -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02_final_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02_final_prof/final_results.db > gpurun_out/r02_final_kernel_stats.txt 2>&1; head -3 gpurun_out/r02_final_kernel_stats.txt | cut -c1-200
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r02_final_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02_final_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02_final_prof_serial/serial_results.db > gpurun_out/r02_final_kernel_stats_serial.txt 2>&1; head -3 gpurun_out/r02_final_kernel_stats_serial.txt | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r02_final_prof_vit -o vit -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 --backbone vits_dino > gpurun_out/r02_final_prof_vit.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02_final_prof_vit/vit_results.db > gpurun_out/r02_final_vit_kernel_stats.txt 2>&1; head -2 gpurun_out/r02_final_vit_kernel_stats.txt | cut -c1-200
