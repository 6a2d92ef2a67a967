# round 3, call 37: conv_wgrad_kernel with two K steps in flight (LP_WGRAD_DEEP, default on) vs one (build/liblp_hip_wd0.so); pool ZU=2 default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv.py tests/test_emu_conv_pipe.py tests/test_emu_vit_ops.py tests/test_emu_trunk_ops.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r03ah_pytest.log; cat gpurun_out/r03ah_pytest.log
for rep in 1 2; do
  for v in deep wd0; do
    lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v = wd0 ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_wd0.so
    LP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16 2>/dev/null | python -c "import sys,json; [print('$v', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done > gpurun_out/r03ah_wgrad_deep.txt 2>&1; cat gpurun_out/r03ah_wgrad_deep.txt
LP_DUMP_LAUNCHES=gpurun_out/r03ah_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03ah_bench_layers.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03ah_launches.json > gpurun_out/r03ah_layer_table.txt 2>&1; tail -1 gpurun_out/r03ah_layer_table.txt
for v in deep wd0; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v = wd0 ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_wd0.so
  LP_HIP_LIB=$lib timeout 300 python bench.py --backbone vits_dino --no-cpu-baseline --no-profile --no-secondary --steps 12 2>/dev/null | python -c "import sys,json; [print('vit $v', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done >> gpurun_out/r03ah_wgrad_deep.txt 2>&1; tail -2 gpurun_out/r03ah_wgrad_deep.txt
