# round 3, call 27: bn_finalize folded into the element-wise pass (lp_bn_apply_fin): device tests, step parity, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_emu_tracker.py tests/test_step_parity.py tests/test_gpu_fullsize.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r03z_pytest.log; tail -2 gpurun_out/r03z_pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16"
for i in 1 2 3; do
  for f in 0 1; do echo -n "fused_finalize=$f "; LP_BN_FUSED_FINALIZE=$f $B 2>&1 | tail -1 | cut -c88-110; done
done > gpurun_out/r03z_bn_fin.txt 2>&1; cat gpurun_out/r03z_bn_fin.txt
for f in 0 1; do echo -n "256px fused_finalize=$f "; LP_BN_FUSED_FINALIZE=$f $B --size 256 2>&1 | tail -1 | cut -c88-110; done
