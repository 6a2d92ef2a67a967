# round 3, call 14: step parity with the tail rules for the full-batch fixtures; fragment-prefetch depth A/B; new bench secondary lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_step_parity.py -q -m gpu --timeout 600 -p no:cacheprovider -s 2>&1 | grep -E "PARITY|passed|failed|Error|assert" | cut -c1-500) > gpurun_out/r03n_step_parity.log; grep -E "passed|failed|Error" gpurun_out/r03n_step_parity.log | tail -8
S="l2.c2:192:48:128:128:3:1:1 l3.c1:192:24:1024:256:1:1:0 l3.c2:192:24:256:256:3:1:1 l3.c3:192:24:256:1024:1:1:0 l4.c2:192:12:512:512:3:1:1 l1.c3:192:96:64:256:1:1:0"
for rep in 1 2; do for v in base frag2; do
  lib=$GRAFT_REPO_ROOT/build/liblp_hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so
  echo "== $v"; LP_HIP_LIB=$lib KINDS=fwd timeout 120 python profiles/conv_layer_bench.py 5 $S 2>&1 | grep fwd
done; done > gpurun_out/r03n_fragsets.txt 2>&1; cat gpurun_out/r03n_fragsets.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12"
for i in 1 2; do
  $B 2>&1 | tail -1 | cut -c80-160
  LP_HIP_LIB=$GRAFT_REPO_ROOT/build/liblp_hip_frag2.so $B 2>&1 | tail -1 | cut -c80-160
done
$B --unfrozen > gpurun_out/r03n_bench_unfrozen.json.log 2>&1; tail -1 gpurun_out/r03n_bench_unfrozen.json.log | cut -c80-160
$B --peaked > gpurun_out/r03n_bench_peaked.json.log 2>&1; tail -1 gpurun_out/r03n_bench_peaked.json.log | cut -c80-160; tail -1 gpurun_out/r03n_bench_peaked.json.log | grep -o '"decode_prune": "[^"]*"'
LP_DECODE_PRUNE=0 $B --peaked > gpurun_out/r03n_bench_peaked_noprune.json.log 2>&1; tail -1 gpurun_out/r03n_bench_peaked_noprune.json.log | cut -c80-160
