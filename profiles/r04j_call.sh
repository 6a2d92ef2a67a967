# Round 4, call J: direct-to-LDS operands vs register-staged operands in the tile loop (profiles/probe/loop_probe.hip, LOOP_PROBE_CALL_D)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/loop_probe profiles/probe/loop_probe.hip 2> gpurun_out/r04j_loop_probe_build.log
LOOP_PROBE_CALL_D=1 timeout 300 /tmp/loop_probe > gpurun_out/r04j_loop_probe.txt 2>&1; cat gpurun_out/r04j_loop_probe.txt
