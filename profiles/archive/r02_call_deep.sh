# forward kernel with two K steps in flight (BN = 64 instantiation): device check + per-layer table + bench in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu -k "emu_conv or ragged or emu_engine or fullsize or step_parity" --timeout 300 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r02s_pytest.log; tail -2 gpurun_out/r02s_pytest.log
LP_DUMP_LAUNCHES=gpurun_out/r02s_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r02s_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r02s_launches.json > gpurun_out/r02s_layer_table.txt 2>&1; head -14 gpurun_out/r02s_layer_table.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02s_bench_$i.json.log 2>&1; tail -1 gpurun_out/r02s_bench_$i.json.log | cut -c1-200; done
timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --size 256 > gpurun_out/r02s_bench_256.json.log 2>&1; tail -1 gpurun_out/r02s_bench_256.json.log | cut -c1-200
