# Round 4, call A: go / no-go probe for the two-workgroup tile + baseline bench of the tree as round 3 left it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/tile_probe profiles/probe/tile_probe.hip 2> gpurun_out/r04a_tile_probe_build.log
timeout 120 /tmp/tile_probe > gpurun_out/r04a_tile_probe.txt 2>&1; cat gpurun_out/r04a_tile_probe.txt
timeout 600 python bench.py > gpurun_out/r04a_bench_n1.json.log 2>&1; tail -1 gpurun_out/r04a_bench_n1.json.log | cut -c1-400
