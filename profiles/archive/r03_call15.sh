# round 3, call 15: HALO form of the 3x3 layers (parity on real shapes, per-layer table, step A/B), Linear layers of the ViT on the pipelined kernel (A/B),
# fp32 validation executor of the ViT (step parity c4 in both precisions)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_emu_vit_engine.py tests/test_emu_conv_pipe.py -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r03o_pytest_halo.log; tail -3 gpurun_out/r03o_pytest_halo.log
(timeout 900 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -s -k "c4 or c2full or c1" 2>&1 | grep -E "PARITY|passed|failed|Error|assert" | cut -c1-400) > gpurun_out/r03o_step_parity.log; grep -E "passed|failed|Error" gpurun_out/r03o_step_parity.log | tail -8
S="l1.c2:192:96:64:64:3:1:1 l2.c2:192:48:128:128:3:1:1 l3.c2:192:24:256:256:3:1:1 l4.c2:192:12:512:512:3:1:1"
for rep in 1 2; do for h in 0 1; do
  echo "== LP_CONV_HALO=$h"; LP_CONV_HALO=$h KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd
done; done > gpurun_out/r03o_halo_layers.txt 2>&1; cat gpurun_out/r03o_halo_layers.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  LP_CONV_HALO=0 $B 2>&1 | tail -1 | cut -c80-160
  LP_CONV_HALO=1 $B 2>&1 | tail -1 | cut -c80-160
done
LP_DUMP_LAUNCHES=gpurun_out/r03o_launches.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r03o_bench_profile.json.log 2>&1
python profiles/layer_table.py gpurun_out/r03o_launches.json > gpurun_out/r03o_layer_table.txt 2>&1; tail -1 gpurun_out/r03o_layer_table.txt
for i in 1 2; do
  LP_GEMM_PIPE=0 $B --backbone vits_dino 2>&1 | tail -1 | cut -c80-160
  LP_GEMM_PIPE=1 $B --backbone vits_dino 2>&1 | tail -1 | cut -c80-160
done
