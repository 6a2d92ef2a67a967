# round 3, call 3: does starting every tile's K walk at another step (no two workgroups on the same 128-B column at once) lift the deep layers?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r03c_bench_$tag.json.log 2>&1; echo "$tag: $(tail -1 gpurun_out/r03c_bench_$tag.json.log | cut -c80-160)"; }
for i in 1 2; do
  run base_$i LP_PIPE_FLAGS=0
  run rot_$i LP_PIPE_FLAGS=4
  run rotnt_$i LP_PIPE_FLAGS=6
done
LP_PIPE_FLAGS=4 LP_DUMP_LAUNCHES=gpurun_out/r03c_launches_rot.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03c_bench_profile_rot.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03c_launches_rot.json > gpurun_out/r03c_layer_table_rot.txt 2>&1; tail -1 gpurun_out/r03c_layer_table_rot.txt
LP_PIPE_FLAGS=0 LP_DUMP_LAUNCHES=gpurun_out/r03c_launches_base.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03c_bench_profile_base.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03c_launches_base.json > gpurun_out/r03c_layer_table_base.txt 2>&1; tail -1 gpurun_out/r03c_layer_table_base.txt
