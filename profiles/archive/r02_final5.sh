# the device suite after the last test additions (ragged loss shapes, non-square frames, ViT token counts), twice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider --tb=short > /tmp/suite$i.log 2>&1; tail -1 /tmp/suite$i.log; (grep -E "^(FAILED|E  )" /tmp/suite$i.log | head -40; tail -3 /tmp/suite$i.log) | cut -c1-300 > gpurun_out/r02_final_pytest_gpu_run$i.log; done
