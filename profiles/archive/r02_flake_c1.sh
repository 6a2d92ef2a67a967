# flake hunt: every device-only step-parity case 12 times (bf16-mixed: run-to-run variation comes from the atomics of small launches), messages kept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; : > gpurun_out/r02_flake_step_parity.log
for i in $(seq 1 12); do timeout 300 python -m pytest tests/test_step_parity.py -q -m gpu -p no:cacheprovider -s -k "bf16 or c4" 2>&1 | grep -E "PARITY|AssertionError|^E  |passed|failed" | cut -c1-500 >> gpurun_out/r02_flake_step_parity.log; done
grep -c "passed" gpurun_out/r02_flake_step_parity.log; grep -c failed gpurun_out/r02_flake_step_parity.log
