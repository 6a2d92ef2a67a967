# round 5, call E: the hybrid form of decode_bwd_kernel (plain ds = 2 kernels: column taps in registers; pruning kernels: re-read; row groups two
# output rows at a time) against the round-4 form rebuilt on the new ABI (build/liblp_hip_decold.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in default decold default decold; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$tag.so; [ $tag = default ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $tag" | tee -a gpurun_out/r05e_decode_variants.txt
  LP_HIP_LIB=$lib timeout 300 python profiles/decode_microbench.py 2>&1 | tail -4 | cut -c1-150 | tee -a gpurun_out/r05e_decode_variants.txt
done
