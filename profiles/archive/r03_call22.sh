cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do (timeout 600 python -m pytest "tests/test_step_parity.py::test_step_parity_baseline_configs[c2full-bf16-mixed]" -q -m gpu --timeout 600 -p no:cacheprovider -s 2>&1 | grep -E "PARITY|passed|failed|Error:|^E " | cut -c1-600); done > gpurun_out/r03u_c2full.log 2>&1; cat gpurun_out/r03u_c2full.log
