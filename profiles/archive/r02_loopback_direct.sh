# RCCL loop-back A/B on the 1-GPU box: where the data-parallel path's device time goes (world of one rank: the collectives move nothing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_widen_bench_helpers.py -q -m gpu -k loopback --timeout 500 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r02_loopback_pytest.log; tail -3 gpurun_out/r02_loopback_pytest.log
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
B="timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 12"
$B > gpurun_out/r02p_solo.json.log 2>&1
export LP_DIST_LOOPBACK=1
MASTER_PORT=29621 $B > gpurun_out/r02p_loop_pg.json.log 2>&1
MASTER_PORT=29622 $B --no-sync-bn > gpurun_out/r02p_loop_nosyncbn.json.log 2>&1
MASTER_PORT=29623 LP_SYNCBN_DIRECT=1 $B > gpurun_out/r02p_loop_direct.json.log 2>&1
MASTER_PORT=29624 LP_SYNCBN_DIRECT=1 LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 $B > gpurun_out/r02p_loop_direct_graph.json.log 2>&1
MASTER_PORT=29625 LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 $B > gpurun_out/r02p_loop_pg_graph.json.log 2>&1
unset LP_DIST_LOOPBACK
$B > gpurun_out/r02p_solo_b.json.log 2>&1
