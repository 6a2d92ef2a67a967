# round 5, call G: the forward store pass's LDS corner - XOR-swizzled unpadded rows (new) vs the padded rows of rounds 3 - 4
# (build/liblp_hip_padcorner.so): alternating bench processes + the SQ counters of both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py tests/test_emu_conv.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3) | tee gpurun_out/r05g_pytest.log
B="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 3"
for rep in 1 2 3; do for tag in new pad; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = pad ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_padcorner.so
  echo "$tag rep $rep: $(LP_HIP_LIB=$lib timeout 300 $B 2>&1 | tail -1 | cut -c80-125)" | tee -a gpurun_out/r05g_corner_ab.txt
done; done
for tag in new pad; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = pad ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_padcorner.so
  LP_HIP_LIB=$lib LP_DUMP_LAUNCHES=gpurun_out/r05g_launches_$tag.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 > gpurun_out/r05g_bench_$tag.json.log 2>&1
  python profiles/layer_table.py gpurun_out/r05g_launches_$tag.json > gpurun_out/r05g_layer_table_$tag.txt 2>&1; echo "$tag $(tail -1 gpurun_out/r05g_layer_table_$tag.txt)" | tee -a gpurun_out/r05g_corner_ab.txt
  LP_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d /tmp/r05g_sq_$tag -o sq -- python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 2 --warmup 1 > gpurun_out/r05g_sq_$tag.log 2>&1
  python profiles/summarize_pmc_sq.py /tmp/r05g_sq_$tag/sq_results.db > gpurun_out/r05g_pmc_mfma_$tag.json 2>/dev/null
done
