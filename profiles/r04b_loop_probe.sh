# Round 4, call B: what bounds conv_pipe_kernel's tile loop (profiles/probe/loop_probe.hip): loop parts, store-pass forms, phase stagger
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/loop_probe profiles/probe/loop_probe.hip 2> gpurun_out/r04b_loop_probe_build.log
timeout 300 /tmp/loop_probe > gpurun_out/r04b_loop_probe.txt 2>&1; cat gpurun_out/r04b_loop_probe.txt
