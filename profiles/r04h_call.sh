# Round 4, call H: A/B of the HALO swizzle key (u = row - 2 x image rows, round 4) against round 3's (the row itself), per layer and per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in new old new old; do
  lib=""; [ $tag = old ] && lib="LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_halorow.so"
  env $lib LP_DUMP_LAUNCHES=gpurun_out/r04h_launches_$tag.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r04h_bench_${tag}_prof.json.log 2>&1
  env $lib timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 10 > gpurun_out/r04h_bench_${tag}.json.log 2>&1
  echo $tag $(tail -1 gpurun_out/r04h_bench_${tag}_prof.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['mfma_ms_per_step'], d['roofline']['frac'])") $(tail -1 gpurun_out/r04h_bench_${tag}.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
done
python profiles/layer_table.py gpurun_out/r04h_launches_new.json > gpurun_out/r04h_layer_table_new.txt 2>&1
python profiles/layer_table.py gpurun_out/r04h_launches_old.json > gpurun_out/r04h_layer_table_old.txt 2>&1
paste <(grep "c2 " gpurun_out/r04h_layer_table_old.txt | cut -c1-75) <(grep "c2 " gpurun_out/r04h_layer_table_new.txt | cut -c20-75)
tail -1 gpurun_out/r04h_layer_table_old.txt; tail -1 gpurun_out/r04h_layer_table_new.txt
