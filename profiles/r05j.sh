# round 5, call J: non-temporal stores in the BatchNorm walks (build/liblp_hip_bnnt.so) - microbench at the real shapes, then the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in default bnnt default bnnt; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = bnnt ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_bnnt.so
  echo "== $tag" | tee -a gpurun_out/r05j_bn_nt.txt
  LP_HIP_LIB=$lib timeout 300 python profiles/bn_microbench.py 2>&1 | tail -6 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][:28].ljust(28), ' '.join(f'{k[:22]}={v[0]}us/{v[1]}' for k, v in d.items() if isinstance(v, list)))" | tee -a gpurun_out/r05j_bn_nt.txt
done
B="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 3"
for rep in 1 2 3; do for tag in default bnnt; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = bnnt ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_bnnt.so
  echo "$tag rep $rep: $(LP_HIP_LIB=$lib timeout 300 $B 2>&1 | tail -1 | cut -c80-125)" | tee -a gpurun_out/r05j_bn_nt.txt
done; done
