# round 2, first device call: the whole -m gpu suite WITHOUT -x (everything written after round 1's budget gets its first device run), smoke,
# the default bench, and the A/Bs of the two opt-in tile-wave remedies.
#   gpurun --timeout 1500 -- 'bash profiles/r02_call1.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 800 python -m pytest tests -q -m gpu --timeout 240 -p no:cacheprovider 2>&1 | tail -150) > gpurun_out/r02_pytest_gpu.log; tail -3 gpurun_out/r02_pytest_gpu.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/r02_smoke.log; tail -1 gpurun_out/r02_smoke.log
timeout 400 python bench.py > gpurun_out/r02_bench_n1.json.log 2>&1; tail -1 gpurun_out/r02_bench_n1.json.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_bench_n1_noprofile.json.log 2>&1; tail -1 gpurun_out/r02_bench_n1_noprofile.json.log | cut -c1-200
LP_CONV_TAIL_BN64=1 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_bench_n1_tail_bn64.json.log 2>&1; tail -1 gpurun_out/r02_bench_n1_tail_bn64.json.log | cut -c1-200
LP_TWO_STREAMS=1 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02_bench_n1_two_streams.json.log 2>&1; tail -1 gpurun_out/r02_bench_n1_two_streams.json.log | cut -c1-200
for bb in resnet50 vits_dino; do
  timeout 300 python bench.py --predict --backbone $bb --steps 10 --warmup 3 > gpurun_out/r02_bench_predict_${bb}.json.log 2>&1; tail -1 gpurun_out/r02_bench_predict_${bb}.json.log | cut -c1-200
done
timeout 300 python bench.py --views 4 --size 256 --labeled 16 --unlabeled 32 --no-cpu-baseline > gpurun_out/r02_bench_c5_multiview.json.log 2>&1; tail -1 gpurun_out/r02_bench_c5_multiview.json.log | cut -c1-200
timeout 300 python bench.py --size 256 --no-cpu-baseline > gpurun_out/r02_bench_256.json.log 2>&1; tail -1 gpurun_out/r02_bench_256.json.log | cut -c1-200
timeout 200 python profiles/producer_microbench.py > gpurun_out/r02_producer_microbench.txt 2>&1; tail -12 gpurun_out/r02_producer_microbench.txt
