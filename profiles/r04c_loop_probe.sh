# Round 4, call C: is conv_pipe_kernel's loop bound by the BYTES through the L2 -> LDS path, the load instructions, or their latency against the ring depth?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/loop_probe profiles/probe/loop_probe.hip 2> gpurun_out/r04c_loop_probe_build.log
timeout 300 /tmp/loop_probe > gpurun_out/r04c_loop_probe.txt 2>&1; cat gpurun_out/r04c_loop_probe.txt
