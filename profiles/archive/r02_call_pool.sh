# stem pool kernels as row walks (no 64-bit divisions per element), coalesced soft-max backward stores: device check + per-kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu -k "trunk_ops or ragged or emu_engine or segmented or softmax or head or step_parity or fullsize" --timeout 300 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r02q_pytest.log; tail -2 gpurun_out/r02q_pytest.log
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r02q_prof -o q -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02q_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02q_prof/q_results.db > gpurun_out/r02q_kernel_stats_serial.txt 2>&1; grep -E "pool|softmax2d|total kernel" gpurun_out/r02q_kernel_stats_serial.txt | cut -c1-70,113-180
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02q_bench_$i.json.log 2>&1; tail -1 gpurun_out/r02q_bench_$i.json.log | cut -c1-200; done
