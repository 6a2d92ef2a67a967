# round 3, call 19: kernel trace of the ViT-S/16 step (config C4) on the current tree: where do its 50 ms go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r03r_vit -o vit -- python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2 > gpurun_out/r03r_vit_prof.log 2>&1
python profiles/summarize_rocpd.py /tmp/r03r_vit/vit_results.db > gpurun_out/r03r_vit_kernel_stats.txt 2>&1; head -45 gpurun_out/r03r_vit_kernel_stats.txt | cut -c1-180
tail -1 gpurun_out/r03r_vit_prof.log | cut -c1-300
