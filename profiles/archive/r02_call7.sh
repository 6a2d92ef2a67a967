cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "LP_CONV_STREAM_OUT=0 LP_BN_VARIANT=0" "LP_CONV_STREAM_OUT=0" "LP_BN_VARIANT=0" ""; do
  tag=$(echo "$v" | tr ' =' '__'); env $v timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02g_bench_${tag:-default}.json.log 2>&1; echo "$v"; tail -1 gpurun_out/r02g_bench_${tag:-default}.json.log | cut -c1-160
done
