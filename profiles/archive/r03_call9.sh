# round 3, call 9: kernel trace of the current tree (serialised: weight gradients on the main stream, as the per-launch events see them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03i_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r03i_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03i_prof_serial/serial_results.db > gpurun_out/r03i_kernel_stats_serial.txt 2>&1; head -40 gpurun_out/r03i_kernel_stats_serial.txt | cut -c1-175
