#!/bin/bash
# r06h: forward moments updated once per row GROUP (es, ew) instead of per row: decode tests, microbench against the corner build (which takes the same change)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_decode.py tests/test_step_parity.py tests/test_differential_vs_reference.py -q -m gpu -x -p no:cacheprovider -k "decode or repeats or c2full or subpixel" 2>&1 | tail -3) | tee gpurun_out/r06h_pytest.txt
for i in 1 2; do
  for lib in nocenter new; do
    if [ $lib = nocenter ]; then export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_nocenter.so; else unset LP_HIP_LIB; fi
    echo "== decode microbench, library: $lib" | tee -a gpurun_out/r06h_decode_ab.txt
    timeout 300 python profiles/decode_microbench.py 2>/dev/null | cut -c1-140 | tee -a gpurun_out/r06h_decode_ab.txt
  done
done
