# round 5, call B: (1) the step-parity numbers of every fixture on the device (the north-star table of tests/test_step_parity.py is set from
# them); (2) conv_spec_kernel after the consumer-scheduling fix: LP_CONV_SPEC=0 / 1 (forward) / 2 (+ data gradient), per-launch events per
# layer, + the consumers' issue priority raised (build/liblp_hip_prio.so); (3) the [gpu] variants of the tests touched since call A
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_spec.py tests/test_segmented_bn.py tests/test_emu_trunk_ops.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r05b_pytest_gpu.log; tail -2 gpurun_out/r05b_pytest_gpu.log
for s in 0 1 2; do
  LP_CONV_SPEC=$s LP_DUMP_LAUNCHES=gpurun_out/r05b_launches_spec$s.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 > gpurun_out/r05b_bench_spec$s.json.log 2>&1
  tail -1 gpurun_out/r05b_bench_spec$s.json.log | cut -c1-170
  python profiles/layer_table.py gpurun_out/r05b_launches_spec$s.json > gpurun_out/r05b_layer_table_spec$s.txt 2>&1; tail -1 gpurun_out/r05b_layer_table_spec$s.txt
done
LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_prio.so LP_CONV_SPEC=1 LP_DUMP_LAUNCHES=gpurun_out/r05b_launches_spec1_prio.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 > gpurun_out/r05b_bench_spec1_prio.json.log 2>&1
tail -1 gpurun_out/r05b_bench_spec1_prio.json.log | cut -c1-170
python profiles/layer_table.py gpurun_out/r05b_launches_spec1_prio.json > gpurun_out/r05b_layer_table_spec1_prio.txt 2>&1; tail -1 gpurun_out/r05b_layer_table_spec1_prio.txt
timeout 900 python profiles/parity_report.py c1 c2 c5 c5v4 c2full c4 c4full > gpurun_out/r05b_parity_device.jsonl 2> gpurun_out/r05b_parity_device.err; wc -l gpurun_out/r05b_parity_device.jsonl
