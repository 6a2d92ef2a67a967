# round 3, call 24: conv_stem2d_kernel (16 x 16 output tiles, resident filter, two workgroups per CU) - parity on the device, step A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_conv_pipe.py tests/test_emu_conv.py tests/test_gpu_fullsize.py -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -5) > gpurun_out/r03w_pytest.log; tail -3 gpurun_out/r03w_pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_STEM_2D=0 $B 2>&1 | tail -1 | cut -c80-160
  LP_STEM_2D=1 $B 2>&1 | tail -1 | cut -c80-160
done
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03w_prof -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r03w_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03w_prof/serial_results.db > gpurun_out/r03w_kernel_stats_serial.txt 2>&1; head -30 gpurun_out/r03w_kernel_stats_serial.txt | cut -c1-60,110-175
(timeout 600 python -m pytest "tests/test_step_parity.py" -q -m gpu --timeout 600 -p no:cacheprovider -k "c2full or c1 or c5v4" 2>&1 | tail -3)
