# round 5, call F: does the backward overlap better when the BatchNorm backward walk (no LDS since round 5: it can share a CU with a pipelined
# weight gradient, 143 VGPRs x 2 waves per SIMD) leaves room - LP_BN_BWD_WGS_PER_CU 5 / 4 / 3 / 2 - or when the weight-gradient stream is given a
# lower / higher priority than the main stream?  bench.py --no-profile (the side stream is on), 10 steps, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())" | tee gpurun_out/r05f_overlap.txt
B="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 3"
one() { echo "$1: $(env $1 timeout 300 $B 2>&1 | tail -1 | cut -c80-125)" | tee -a gpurun_out/r05f_overlap.txt; }
for rep in 1 2; do
  one LP_BN_BWD_WGS_PER_CU=5
  one LP_BN_BWD_WGS_PER_CU=4
  one LP_BN_BWD_WGS_PER_CU=3
  one LP_BN_BWD_WGS_PER_CU=2
  one LP_WGRAD_STREAM_PRIORITY=-1
  one LP_WGRAD_STREAM_PRIORITY=1
  one LP_WGRAD_SIDE_STREAM=0
done
