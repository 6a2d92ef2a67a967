# round 5, call H: the HALO form's weight ring with FOUR stages (three K steps of loads in flight; 160 KB of LDS) against three (build/liblp_hip_halo3.so):
# is the K step of the 3x3 layers bound by load latency / prefetch depth?  [gpu] bit-identity tests, alternating bench processes, per-layer tables
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_emu_conv_pipe.py tests/test_gpu_fullsize.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -2) | tee gpurun_out/r05h_pytest.log
B="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 10 --warmup 3"
for rep in 1 2 3; do for tag in halo4 halo3; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = halo3 ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_halo3.so
  echo "$tag rep $rep: $(LP_HIP_LIB=$lib timeout 300 $B 2>&1 | tail -1 | cut -c80-125)" | tee -a gpurun_out/r05h_halo_stages.txt
done; done
for tag in halo4 halo3; do
  lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $tag = halo3 ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_halo3.so
  LP_HIP_LIB=$lib LP_DUMP_LAUNCHES=gpurun_out/r05h_launches_$tag.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 > gpurun_out/r05h_bench_$tag.json.log 2>&1
  python profiles/layer_table.py gpurun_out/r05h_launches_$tag.json > gpurun_out/r05h_layer_table_$tag.txt 2>&1; echo "$tag $(tail -1 gpurun_out/r05h_layer_table_$tag.txt)" | tee -a gpurun_out/r05h_halo_stages.txt
done
