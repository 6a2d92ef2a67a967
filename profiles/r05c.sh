# round 5, call C: the whole device suite on the cleaned tree; A/B of the two-launch BatchNorm backward (LP_BN_BWD_TERMS=1 / 0, alternating pairs);
# serialised kernel trace of the default step (per-kernel times of the rewritten kernels: bn_bwd_apply<true>, decode_bwd, softmax2d_bwd_pixmajor, pca)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r05c_pytest_gpu.log; tail -3 gpurun_out/r05c_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
for rep in 1 2 3; do for t in 1 0; do
  LP_BN_BWD_TERMS=$t timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 3 > gpurun_out/r05c_bench_terms${t}_$rep.json.log 2>&1
  echo "terms=$t rep=$rep $(tail -1 gpurun_out/r05c_bench_terms${t}_$rep.json.log | cut -c80-130)"
done; done
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r05c_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 8 --warmup 2 > gpurun_out/r05c_prof_serial.log 2>&1
python profiles/summarize_rocpd.py /tmp/r05c_prof_serial/serial_results.db > gpurun_out/r05c_kernel_stats_serial.txt 2>&1; head -14 gpurun_out/r05c_kernel_stats_serial.txt | cut -c1-60,110-160
