# closing tree after the per-workgroup BatchNorm sums: step parity x3 (run-to-run variation now in every layer), the device suite, a bench line with the per-launch roofline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; : > gpurun_out/r02_final_step_parity_x3.log
for i in 1 2 3; do timeout 200 python -m pytest tests/test_step_parity.py -q -m gpu -p no:cacheprovider -s -k "bf16 or c4" 2>&1 | grep -E "PARITY|^E  |passed|failed" | cut -c1-420 >> gpurun_out/r02_final_step_parity_x3.log; done; grep -E "passed|failed" gpurun_out/r02_final_step_parity_x3.log | tr '\n' ' '; echo
timeout 600 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider --tb=short > /tmp/suite1.log 2>&1; tail -1 /tmp/suite1.log; (grep -E "^(FAILED|E  )" /tmp/suite1.log | head -40; tail -3 /tmp/suite1.log) | cut -c1-300 > gpurun_out/r02_final_pytest_gpu_run4.log
timeout 200 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02_final_bench_n1_wgstats.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_n1_wgstats.json.log | cut -c80-230
