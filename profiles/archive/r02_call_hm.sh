# heat-map generation (separable profiles) on the device + rocprofv3 kernel trace of the RCCL loop-back step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -q -m gpu -k "heatmap or generate or gauss or dataset or bench" --timeout 300 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r02n_pytest_hm.log; tail -2 gpurun_out/r02n_pytest_hm.log
timeout 200 python - > gpurun_out/r02n_hbm_rooflines.json 2>&1 <<'PY'
import json, torch, bench
print(json.dumps(bench.hbm_rooflines(torch.device("cuda:0"), 192, 17, 384), indent=1))
PY
grep -A4 heatmap_gen gpurun_out/r02n_hbm_rooflines.json | head -6
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29621 LP_DIST_LOOPBACK=1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r02n_prof_loop -o loop -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r02n_prof_loop.log 2>&1
python profiles/summarize_rocpd.py /tmp/r02n_prof_loop/loop_results.db > gpurun_out/r02n_loopback_kernel_stats.txt 2>&1; grep -i "nccl\|rccl" gpurun_out/r02n_loopback_kernel_stats.txt | cut -c1-60,100-200 | head
