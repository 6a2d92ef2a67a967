# round 3, call 45: inference (lp_conv_fwd_act) on the pipelined forward kernel vs conv_igemm_kernel<infer> (LP_INFER_PIPE=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_widen_inference_kernels.py tests/test_widen_inference_stack.py tests/test_widen_n3_predictions.py tests/test_emu_conv_pipe.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r03ao_pytest.log; cat gpurun_out/r03ao_pytest.log
for rep in 1 2; do
  for v in 1 0; do
    LP_INFER_PIPE=$v timeout 300 python bench.py --predict --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('LP_INFER_PIPE=$v resnet50', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r03ao_infer_pipe.txt 2>&1
for v in 1 0; do
  LP_INFER_PIPE=$v timeout 300 python bench.py --predict --backbone vits_dino --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('LP_INFER_PIPE=$v vits', d['value'], d['ms_per_step'])"
done >> gpurun_out/r03ao_infer_pipe.txt 2>&1; cat gpurun_out/r03ao_infer_pipe.txt
