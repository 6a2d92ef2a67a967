# round 3, call 34: per-layer table of the final kernels (headline run only: the default bench's secondary configs overwrite the launch dump), ViT trace of the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LP_DUMP_LAUNCHES=gpurun_out/r03_final_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03_final_bench_layers.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03_final_launches.json > gpurun_out/r03_final_layer_table.txt 2>&1; tail -1 gpurun_out/r03_final_layer_table.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03_final_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r03_final_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03_final_vit/vit_results.db > gpurun_out/r03_final_vit_kernel_stats.txt 2>&1; head -16 gpurun_out/r03_final_vit_kernel_stats.txt | cut -c1-60,110-160
