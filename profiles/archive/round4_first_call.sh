# Next round's first gpurun call (written at the end of round 3, after the GPU budget was spent; nothing here has run yet):
#   1. the go / no-go probe for the two-workgroup tile shape: profiles/probe/tile_probe.hip
#   2. the device suite and the default bench on the tree as round 3 left it (baseline for the round's A/B runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/tile_probe profiles/probe/tile_probe.hip 2> gpurun_out/r04a_tile_probe_build.log
timeout 120 /tmp/tile_probe > gpurun_out/r04a_tile_probe.txt 2>&1; cat gpurun_out/r04a_tile_probe.txt
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r04a_pytest_gpu.log; tail -2 gpurun_out/r04a_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r04a_bench_n1.json.log 2>&1; tail -1 gpurun_out/r04a_bench_n1.json.log | cut -c1-300
