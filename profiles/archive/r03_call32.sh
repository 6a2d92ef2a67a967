# round 3, call 32: the device suite three more times on the final tree (flake hunt: the driver runs it once with -x)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do (timeout 900 python -m pytest tests -x -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r03_final_pytest_gpu_run$i.log; tail -1 gpurun_out/r03_final_pytest_gpu_run$i.log; done
