# round 4, call m: where the fixed-point build loses against the slot-row build - kernel traces of both (serialised: weight gradients in line)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --no-cpu-baseline --no-profile --no-secondary --steps 6 --warmup 2"
(cd build/ab_slots && LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04m_slots -o t -- python $B > /dev/null 2>&1)
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r04m_fx -o t -- python $B > /dev/null 2>&1
python profiles/summarize_rocpd.py /tmp/r04m_slots/t_results.db > gpurun_out/r04m_kernel_stats_slots.txt 2>&1
python profiles/summarize_rocpd.py /tmp/r04m_fx/t_results.db > gpurun_out/r04m_kernel_stats_fx.txt 2>&1
head -30 gpurun_out/r04m_kernel_stats_fx.txt | cut -c1-70,100-150
