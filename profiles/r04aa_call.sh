# round 4, call aa: conv_wgrad_spec_kernel (4 consumer + 4 producer waves) vs conv_wgrad_pipe_kernel<128> (8 symmetric waves, LP_WGRAD_SPEC=0):
# bit-identity on the device, three alternating pairs of the step, per-kernel times (serialised trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r04aa_pytest_gpu.log; tail -1 gpurun_out/r04aa_pytest_gpu.log
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3"
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  LP_WGRAD_SPEC=0 timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04aa_bench_sym_$i.json.log
  timeout 300 python $B 2>&1 | tail -1 > gpurun_out/r04aa_bench_spec_$i.json.log
  echo "pair $i: symmetric $(v gpurun_out/r04aa_bench_sym_$i.json.log) specialised $(v gpurun_out/r04aa_bench_spec_$i.json.log)"
done
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
LP_WGRAD_SPEC=0 LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04aa_sym -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04aa_sym/t_results.db > gpurun_out/r04aa_kernel_stats_sym.txt 2>&1
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04aa_spec -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04aa_spec/t_results.db > gpurun_out/r04aa_kernel_stats_spec.txt 2>&1
grep -h "conv_wgrad_pipe_kernel<128>\|conv_wgrad_spec" gpurun_out/r04aa_kernel_stats_sym.txt gpurun_out/r04aa_kernel_stats_spec.txt | cut -c1-50,105-175
