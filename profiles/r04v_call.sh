# round 4, call v: the device suite + smoke on the final tree after the profiles/archive move (tests read profiles/archive/r03_policy_grad_c2full.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/r04_final_pytest_gpu.log; tail -2 gpurun_out/r04_final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-200
