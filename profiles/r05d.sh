# round 5, call D: which form of decode_bwd_kernel's row-group loop (decode.hip: LP_DEC_RR_FULL / LP_DEC_RR_PART / LP_DEC_TX_REGS)?
# profiles/decode_microbench.py (192 x 17 maps of 96 x 96, flat and peaked, plain and pruned kernels) on four builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in default decold dec361 dec162; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$tag.so; [ $tag = default ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $tag" | tee -a gpurun_out/r05d_decode_variants.txt
  LP_HIP_LIB=$lib timeout 300 python profiles/decode_microbench.py 2>&1 | tail -4 | tee -a gpurun_out/r05d_decode_variants.txt
done
