# round 4, call y: measured HBM bytes and rate of the BatchNorm streaming kernels (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 2 --warmup 1"
LP_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r04y_fetch -o fetch -- $CMD > /dev/null 2>&1
LP_WGRAD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r04y_write -o write -- $CMD > /dev/null 2>&1
python profiles/summarize_pmc_bn.py /tmp/r04y_fetch/fetch_results.db /tmp/r04y_write/write_results.db > gpurun_out/r04_pmc_bn_traffic.json 2> gpurun_out/r04_pmc_bn_traffic.err
cat gpurun_out/r04_pmc_bn_traffic.json | head -60; tail -3 gpurun_out/r04_pmc_bn_traffic.err
