# round 3, call 12: whole device suite on the current tree + step parity incl. the regenerated full-batch fixture
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r03l_pytest_gpu.log; tail -25 gpurun_out/r03l_pytest_gpu.log | cut -c1-300
