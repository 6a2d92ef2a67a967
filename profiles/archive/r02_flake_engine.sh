# flake hunt: the joint + normalise-on-load block-wise engine test on the device until it fails (at most 45 runs), full failure text kept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; : > gpurun_out/r02_flake_engine.log
for i in $(seq 1 45); do
  timeout 120 python -m pytest "tests/test_emu_engine.py::test_engine_forward_backward_blockwise[gpu-joint-norm-on-load]" -q -m gpu -p no:cacheprovider -s > /tmp/one.log 2>&1
  grep -E "MARGINS" /tmp/one.log | cut -c1-300 >> gpurun_out/r02_flake_engine.log
  if grep -q failed /tmp/one.log; then echo "=== FAILED at run $i" >> gpurun_out/r02_flake_engine.log; grep -E "^E |Error|assert" /tmp/one.log | head -40 | cut -c1-300 >> gpurun_out/r02_flake_engine.log; fi
done
grep -c MARGINS gpurun_out/r02_flake_engine.log; grep -c FAILED gpurun_out/r02_flake_engine.log
