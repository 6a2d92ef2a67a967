cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
$B > gpurun_out/r02_final_bench_solo_b.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_solo_b.json.log | cut -c1-200
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 LP_DIST_LOOPBACK=1
MASTER_PORT=29661 $B > gpurun_out/r02_final_bench_loopback_rccl.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_loopback_rccl.json.log | cut -c1-200
MASTER_PORT=29662 $B > gpurun_out/r02_final_bench_loopback_rccl_b.json.log 2>&1; tail -1 gpurun_out/r02_final_bench_loopback_rccl_b.json.log | cut -c1-200
(timeout 300 python -m pytest tests/test_widen_bench_helpers.py -q -m gpu -k loopback -p no:cacheprovider 2>&1 | tail -2)
