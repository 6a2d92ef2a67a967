cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 LP_DIST_LOOPBACK=1
B="timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 12"
for i in 1 2 3; do MASTER_PORT=2963$i LP_SYNCBN_DIRECT=1 LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 $B > gpurun_out/r02p_loop_direct_graph_$i.json.log 2>&1; MASTER_PORT=2964$i LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 $B > gpurun_out/r02p_loop_pg_graph_$i.json.log 2>&1; done
timeout 300 python -m pytest tests/test_graph_step.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
