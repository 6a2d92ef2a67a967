# round 3, call 33: kEkAZB data gradient without its 7 spilled VGPRs (output offsets recomputed): parity on real shapes, per-layer and step A/B vs the previous build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_emu_conv_pipe.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -2)
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16"
for i in 1 2 3; do
  echo -n "head "; LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_head.so $B 2>&1 | tail -1 | cut -c88-110
  echo -n "nospill "; $B 2>&1 | tail -1 | cut -c88-110
done > gpurun_out/r03ae_azb_nospill.txt 2>&1; cat gpurun_out/r03ae_azb_nospill.txt
(timeout 600 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -k "c2full or c1" 2>&1 | tail -2)
