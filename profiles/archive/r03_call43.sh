# round 3, call 43: colreduce_kernel with 1 / 2 / 4 rows per trip (LP_COLREDUCE_U; 4 = the product build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for u in 1 2 4; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $u != 4 ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_cu$u.so
  echo "== LP_COLREDUCE_U=$u"; LP_HIP_LIB=$lib timeout 120 python profiles/colreduce_microbench.py
done > gpurun_out/r03an_colreduce_u.txt 2>&1; cat gpurun_out/r03an_colreduce_u.txt
