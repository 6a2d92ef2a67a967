cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 4 8 16 32; do echo "== LP_POOL_WGS_PER_CU=$w"; LP_POOL_WGS_PER_CU=$w timeout 120 python profiles/pool_microbench.py 128 2>&1 | grep -E "us "; done > gpurun_out/r03ab_pool.txt 2>&1; echo "== default" >> gpurun_out/r03ab_pool.txt; timeout 120 python profiles/pool_microbench.py 128 2>&1 | grep -E "us " >> gpurun_out/r03ab_pool.txt; cat gpurun_out/r03ab_pool.txt
