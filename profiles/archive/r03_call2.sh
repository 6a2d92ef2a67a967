# round 3, call 2: conv_pipe_kernel with (a) loads spread between the k-slices vs one burst, (b) non-temporal output stores, (c) the data
# gradient's read-backs issued a chunk early, full-line store pass - A/B in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py tests/test_segmented_bn.py tests/test_graph_step.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r03b_pytest_pipe.log; tail -3 gpurun_out/r03b_pytest_pipe.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r03b_bench_$tag.json.log 2>&1; echo "$tag: $(tail -1 gpurun_out/r03b_bench_$tag.json.log | cut -c80-160)"; }
for i in 1 2; do
  run spread_$i LP_CONV_PIPE=1
  run nospread_$i LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nospread.so
  run ntstore_$i LP_PIPE_FLAGS=2
  run igemm_$i LP_CONV_PIPE=0
done
LP_DUMP_LAUNCHES=gpurun_out/r03b_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03b_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03b_launches.json > gpurun_out/r03b_layer_table.txt 2>&1; tail -1 gpurun_out/r03b_layer_table.txt
LP_PIPE_FLAGS=2 LP_DUMP_LAUNCHES=gpurun_out/r03b_launches_nt.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03b_bench_profile_nt.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03b_launches_nt.json > gpurun_out/r03b_layer_table_nt.txt 2>&1; tail -1 gpurun_out/r03b_layer_table_nt.txt
