# round 3, call 42: lazy forward store pass (units of the previous tile's store pass between the k-slices of the next tile) vs build/liblp_hip_nolazy.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_emu_vit_ops.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r03am_pytest.log; cat gpurun_out/r03am_pytest.log
for v in lazy nolazy; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v != lazy ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so
  LP_HIP_LIB=$lib LP_DUMP_LAUNCHES=gpurun_out/r03am_launches_$v.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 > gpurun_out/r03am_bench_$v.log 2>&1
  python profiles/layer_table.py gpurun_out/r03am_launches_$v.json > gpurun_out/r03am_layer_table_$v.txt 2>&1; echo $v; tail -1 gpurun_out/r03am_layer_table_$v.txt
done
for rep in 1 2; do
  for v in lazy nolazy; do
    lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v != lazy ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so
    LP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16 2>/dev/null | python -c "import sys,json; [print('$v', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done > gpurun_out/r03am_lazy_ab.txt 2>&1; cat gpurun_out/r03am_lazy_ab.txt
