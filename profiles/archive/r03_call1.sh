# round 3, call 1: the pipelined convolution kernel (conv_pipe.h) on the device - race / equality checks, then A/B against conv_igemm_kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py tests/test_segmented_bn.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r03a_pytest_pipe.log; tail -3 gpurun_out/r03a_pytest_pipe.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_CONV_PIPE=1 $B > gpurun_out/r03a_bench_pipe1_$i.json.log 2>&1; tail -1 gpurun_out/r03a_bench_pipe1_$i.json.log | cut -c80-220
  LP_CONV_PIPE=0 $B > gpurun_out/r03a_bench_pipe0_$i.json.log 2>&1; tail -1 gpurun_out/r03a_bench_pipe0_$i.json.log | cut -c80-220
done
for m in 1 0; do
  LP_CONV_PIPE=$m LP_DUMP_LAUNCHES=gpurun_out/r03a_launches_pipe$m.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03a_bench_profile_pipe$m.json.log 2>&1
  python profiles/layer_table.py gpurun_out/r03a_launches_pipe$m.json > gpurun_out/r03a_layer_table_pipe$m.txt 2>&1; tail -1 gpurun_out/r03a_layer_table_pipe$m.txt
done
LP_CONV_PIPE=1 $B --size 256 > gpurun_out/r03a_bench_pipe1_256.json.log 2>&1; tail -1 gpurun_out/r03a_bench_pipe1_256.json.log | cut -c80-220
LP_CONV_PIPE=0 $B --size 256 > gpurun_out/r03a_bench_pipe0_256.json.log 2>&1; tail -1 gpurun_out/r03a_bench_pipe0_256.json.log | cut -c80-220
