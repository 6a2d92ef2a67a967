# round 4, call x: Trainer.fit line with the labeled images staged on a copy stream (HostStager) - fit vs bare step on one box; data-path tests on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_widen_n2_datamodule.py tests/test_widen_n1n2_stack.py tests/test_widen_n1n2_kernels.py tests/test_widen_bench_helpers.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r04x_pytest_gpu.log; tail -1 gpurun_out/r04x_pytest_gpu.log
v() { grep -o '"value": [0-9.]*' $1 | head -1 | cut -c10-; }
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3 2>&1 | tail -1 > gpurun_out/r04x_bench_step_$i.json.log
  timeout 300 python bench.py --fit --no-cpu-baseline --no-profile --no-secondary --steps 12 --warmup 3 2>&1 | tail -1 > gpurun_out/r04x_bench_fit_$i.json.log
  echo "round $i: step $(v gpurun_out/r04x_bench_step_$i.json.log) fit $(v gpurun_out/r04x_bench_fit_$i.json.log)"
done
