# round 3, call 30: the stem's backward pool passes with dy / arg-max rows staged in LDS (bn_pool_bwd_v2_kernel): microbench A/B, device tests, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do echo "== LP_POOL_V2=$v"; LP_POOL_V2=$v timeout 120 python profiles/pool_microbench.py 128 2>&1 | grep -E "us "; done > gpurun_out/r03ac_pool_v2.txt 2>&1; cat gpurun_out/r03ac_pool_v2.txt
(timeout 600 python -m pytest tests/test_emu_trunk_ops.py tests/test_emu_ragged_shapes.py tests/test_emu_engine.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -2)
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16"
for i in 1 2; do for v in 0 1; do echo -n "pool_v2=$v "; LP_POOL_V2=$v $B 2>&1 | tail -1 | cut -c88-110; done; done
