#!/bin/bash
# r06r: where Trainer.fit's 6 % go (bench.py --fit: pinned uint8 host frames -> device-side producers -> the step): kernel trace + memory-copy trace of the fit line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --fit --steps 12 --warmup 3 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r06r_fit_line.txt
rm -rf /tmp/r06r_prof
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/r06r_prof -o fit -- python bench.py --fit --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --no-profile > /dev/null 2>&1
python profiles/summarize_rocpd.py $(ls /tmp/r06r_prof/*results.db /tmp/r06r_prof/*/*results.db 2>/dev/null | head -1) > gpurun_out/r06r_fit_kernel_stats.txt 2>&1
grep -v "lp::conv\|lp::bn_\|wgrad" gpurun_out/r06r_fit_kernel_stats.txt | head -50 | cut -c1-70,100-170
python profiles/gap_analysis.py $(ls /tmp/r06r_prof/*results.db /tmp/r06r_prof/*/*results.db 2>/dev/null | head -1) 2>&1 | head -12 | tee gpurun_out/r06r_fit_gap.txt
