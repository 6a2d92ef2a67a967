#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in new smprobe1 smprobe2; do
  if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
  rm -rf /tmp/r06v_prof
  LP_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r06v_prof -o t -- python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
  python profiles/summarize_rocpd.py $(ls /tmp/r06v_prof/*results.db /tmp/r06v_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "softmax2d_bwd" | cut -c1-70,100-170 | sed "s/^/$lib /" | tee -a gpurun_out/r06v_softmax_probe.txt
done
