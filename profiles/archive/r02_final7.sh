# rocprofv3 kernel trace of the closing tree: the bench command with the weight gradients on their side stream (default) and serialised
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r02_final_prof -o final -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02_final_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02_final_prof/final_results.db > gpurun_out/r02_final_kernel_stats.txt 2>&1; head -3 gpurun_out/r02_final_kernel_stats.txt | cut -c1-200
LP_WGRAD_SIDE_STREAM=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r02_final_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02_final_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02_final_prof_serial/serial_results.db > gpurun_out/r02_final_kernel_stats_serial.txt 2>&1; head -3 gpurun_out/r02_final_kernel_stats_serial.txt | cut -c1-200
