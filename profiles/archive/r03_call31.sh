cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 3 4 6 8 12; do echo "== LP_POOL_BAND=$b"; LP_POOL_BAND=$b timeout 120 python profiles/pool_microbench.py 128 2>&1 | grep -E "bwd"; done > gpurun_out/r03ad_pool_band.txt 2>&1; cat gpurun_out/r03ad_pool_band.txt
echo "== B=64"; timeout 120 python profiles/pool_microbench.py 64 2>&1 | grep -E "us "
