# round 3, call 23: conv_res2d_kernel (layer1 conv2: 16 x 16 tiles, filter resident in LDS) - parity on the device, per-layer and step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_emu_conv_pipe.py -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -8) > gpurun_out/r03v_pytest.log; tail -3 gpurun_out/r03v_pytest.log
S="l1.c2:192:96:64:64:3:1:1 l1.c2s:192:64:64:64:3:1:1"
for rep in 1 2; do for h in 0 1; do
  echo "== LP_CONV_RES2D=$h"; LP_CONV_RES2D=$h KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd
done; done > gpurun_out/r03v_res2d_layers.txt 2>&1; cat gpurun_out/r03v_res2d_layers.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_CONV_RES2D=0 $B 2>&1 | tail -1 | cut -c80-160
  LP_CONV_RES2D=1 $B 2>&1 | tail -1 | cut -c80-160
done
LP_DUMP_LAUNCHES=gpurun_out/r03v_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03v_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03v_launches.json > gpurun_out/r03v_layer_table.txt 2>&1; head -12 gpurun_out/r03v_layer_table.txt; tail -1 gpurun_out/r03v_layer_table.txt
(timeout 600 python -m pytest "tests/test_step_parity.py" -q -m gpu --timeout 600 -p no:cacheprovider -k "c2full or c1" 2>&1 | tail -3)
