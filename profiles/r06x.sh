#!/bin/bash
# r06x: timing builds of attn_fwd_kernel (-DLP_ATTN_PROBE bits: 1 no P store, 2 no O store, 4 pass 1 over ONE key tile, 8 no exponentials in pass 2, 15 all):
# what the kernel's 0.40 ms per layer are made of (ViT-S/16 step traces; the numerics of these builds are wrong by construction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in new attnp1 attnp2 attnp4 attnp8 attnp15; do
  if [ $lib = new ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_$lib.so; fi
  rm -rf /tmp/r06x_prof
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r06x_prof -o t -- python bench.py --backbone vits_dino --steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
  python profiles/summarize_rocpd.py $(ls /tmp/r06x_prof/*results.db /tmp/r06x_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "attn_fwd" | cut -c1-50,100-170 | sed "s/^/$lib /" | tee -a gpurun_out/r06x_attn_probe.txt
done
