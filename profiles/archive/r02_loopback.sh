# RCCL loop-back on the 1-GPU box: every collective of the data-parallel step over backend "nccl" on a world of one rank
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_widen_bench_helpers.py -q -m gpu -k loopback --timeout 500 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r02_loopback_pytest.log; tail -3 gpurun_out/r02_loopback_pytest.log
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 LP_DIST_LOOPBACK=1
timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_loopback_rccl_resnet50.json.log 2>&1; tail -1 gpurun_out/r02_loopback_rccl_resnet50.json.log | cut -c1-400
MASTER_PORT=29612 timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 --backbone vits_dino > gpurun_out/r02_loopback_rccl_vits.json.log 2>&1; tail -1 gpurun_out/r02_loopback_rccl_vits.json.log | cut -c1-400
MASTER_PORT=29613 LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_loopback_rccl_graph.json.log 2>&1; tail -1 gpurun_out/r02_loopback_rccl_graph.json.log | cut -c1-400
MASTER_PORT=29614 LP_SYNCBN_DIRECT=1 timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_loopback_rccl_direct.json.log 2>&1; tail -1 gpurun_out/r02_loopback_rccl_direct.json.log | cut -c1-400
MASTER_PORT=29615 LP_SYNCBN_DIRECT=1 LP_HIP_GRAPH=1 LP_HIP_GRAPH_DIST=1 timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_loopback_rccl_direct_graph.json.log 2>&1; tail -1 gpurun_out/r02_loopback_rccl_direct_graph.json.log | cut -c1-400
unset LP_DIST_LOOPBACK; timeout 400 python bench.py --no-secondary --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_loopback_solo.json.log 2>&1
