#!/bin/bash
# r05n: where the main stream waits for the side stream (kernel trace of the default step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace -d /tmp/r05n_prof -o t -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r05n_prof.log 2>&1
python profiles/stream_tail.py /tmp/r05n_prof/t_results.db > gpurun_out/r05n_stream_tail.txt 2>&1
cat gpurun_out/r05n_stream_tail.txt
