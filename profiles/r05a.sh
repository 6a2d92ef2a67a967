# round 5, call A: (1) the whole device suite on the changed tree (differential tests now run on the device, north-star 1e-2 xfails, decode `prune`
# argument, switch table, pca_kernel, two-launch bn_bwd_apply, conv_spec_kernel's [gpu] variants); (2) A/B of conv_spec_kernel against
# conv_pipe_kernel in the real step: LP_CONV_SPEC=0 / 1 alternating, per-launch events of a sampled step dumped per layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r05a_pytest_gpu.log; tail -3 gpurun_out/r05a_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
for rep in 1 2; do for s in 0 1; do
  LP_CONV_SPEC=$s LP_DUMP_LAUNCHES=gpurun_out/r05a_launches_spec${s}_$rep.json timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 > gpurun_out/r05a_bench_spec${s}_$rep.json.log 2>&1
  tail -1 gpurun_out/r05a_bench_spec${s}_$rep.json.log | cut -c1-200
done; done
for s in 0 1; do python profiles/layer_table.py gpurun_out/r05a_launches_spec${s}_2.json > gpurun_out/r05a_layer_table_spec$s.txt 2>&1; tail -1 gpurun_out/r05a_layer_table_spec$s.txt; done
for s in 0 1; do LP_CONV_SPEC=$s timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r05a_bench_noprofile_spec$s.json.log 2>&1; tail -1 gpurun_out/r05a_bench_noprofile_spec$s.json.log | cut -c1-200; done
