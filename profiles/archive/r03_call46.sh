# round 3, call 46: inference on the pipelined kernel with the residual requested ahead of the store pass; training step with the new store pass vs the previous commit's library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_widen_inference_kernels.py tests/test_widen_inference_stack.py tests/test_emu_conv_pipe.py -m gpu -q 2>&1 | tail -2 > gpurun_out/r03aq_pytest.log; cat gpurun_out/r03aq_pytest.log
for rep in 1 2; do
  for v in 1 0; do
    LP_INFER_PIPE=$v timeout 300 python bench.py --predict --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('predict LP_INFER_PIPE=$v', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r03aq_infer_pipe.txt 2>&1
for rep in 1 2; do
  for v in new prev; do
    lib=$GRAFT_REPO_ROOT/lightning-pose_amd/liblp_hip.so; [ $v = prev ] && lib=$GRAFT_REPO_ROOT/build/liblp_hip_prev.so
    LP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16 2>/dev/null | python -c "import sys,json; [print('train $v', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done >> gpurun_out/r03aq_infer_pipe.txt 2>&1; cat gpurun_out/r03aq_infer_pipe.txt
