# round 3, call 8: weight-gradient plan (split cost model, HBM-bound shapes stay on conv_wgrad_kernel, tracked coordinates for 3x3, offset-field
# transpose reads); step parity at BASELINE's real batch (c2full) and four views (c5v4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py tests/test_segmented_bn.py tests/test_emu_engine.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r03h_pytest.log; tail -3 gpurun_out/r03h_pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
run() { tag=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" $B "$@" > gpurun_out/r03h_bench_$tag.json.log 2>&1; echo "$tag: $(tail -1 gpurun_out/r03h_bench_$tag.json.log | cut -c80-160)"; }
for i in 1 2; do
  run wgpipe_$i LP_WGRAD_PIPE=1
  run wgold_$i LP_WGRAD_PIPE=0
done
LP_DUMP_LAUNCHES=gpurun_out/r03h_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03h_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03h_launches.json > gpurun_out/r03h_layer_table.txt 2>&1; tail -1 gpurun_out/r03h_layer_table.txt
(timeout 900 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -s 2>&1 | grep -E "PARITY|passed|failed|Error|assert" | cut -c1-400) > gpurun_out/r03h_step_parity.log; tail -12 gpurun_out/r03h_step_parity.log
