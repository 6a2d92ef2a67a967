# round 3, call 7 (transpose reads as asm: no vmcnt(0) per k-slice): pipelined weight gradient (conv_wgrad_pipe_kernel) - device tests, A/B against conv_wgrad_kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r03g_pytest.log; tail -3 gpurun_out/r03g_pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
run() { tag=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" $B "$@" > gpurun_out/r03g_bench_$tag.json.log 2>&1; echo "$tag: $(tail -1 gpurun_out/r03g_bench_$tag.json.log | cut -c80-160)"; }
for i in 1 2; do
  run wgpipe_$i LP_WGRAD_PIPE=1
  run wgold_$i LP_WGRAD_PIPE=0
done
LP_DUMP_LAUNCHES=gpurun_out/r03g_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03g_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03g_launches.json > gpurun_out/r03g_layer_table.txt 2>&1; tail -1 gpurun_out/r03g_layer_table.txt
run wgpipe_256 LP_WGRAD_PIPE=1 --size 256
