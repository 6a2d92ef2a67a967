# round 3, call 44: the whole device suite twice with failure details (flake hunt after 3 failures in one closing run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do (timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed|^tests/.*Error" | cut -c1-600) > gpurun_out/r03_suite_detail_$i.log; tail -4 gpurun_out/r03_suite_detail_$i.log; done
