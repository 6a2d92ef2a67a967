cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for zu in 2 3 6; do echo "== LP_POOL_ZU=$zu"; LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_zu$zu.so timeout 120 python profiles/pool_microbench.py 128 2>&1 | grep -E "bwd"; done > gpurun_out/r03ag_pool_zu.txt 2>&1; cat gpurun_out/r03ag_pool_zu.txt
