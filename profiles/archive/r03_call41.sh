# round 3, call 41: forward store pass split - global stores skipped (nogstore) / BatchNorm sums skipped (nostats) / whole pass skipped (nostore); timing only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in prod nogstore nostats nostore; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v != prod ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so
  LP_HIP_LIB=$lib LP_DUMP_LAUNCHES=gpurun_out/r03al_launches_$v.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 > gpurun_out/r03al_bench_$v.log 2>&1
  python profiles/layer_table.py gpurun_out/r03al_launches_$v.json > gpurun_out/r03al_layer_table_$v.txt 2>&1; echo $v; tail -1 gpurun_out/r03al_layer_table_$v.txt
done
