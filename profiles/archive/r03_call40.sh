# round 3, call 40: what the forward store pass costs - per-layer forward times with the pass skipped (build/liblp_hip_nostore.so, results invalid) vs the product
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in prod nostore; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v = nostore ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_nostore.so
  LP_HIP_LIB=$lib LP_DUMP_LAUNCHES=gpurun_out/r03ak_launches_$v.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 > gpurun_out/r03ak_bench_$v.log 2>&1
  python profiles/layer_table.py gpurun_out/r03ak_launches_$v.json > gpurun_out/r03ak_layer_table_$v.txt 2>&1; tail -1 gpurun_out/r03ak_layer_table_$v.txt
done
