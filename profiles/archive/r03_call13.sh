cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python profiles/parity_dist.py c5v4 bf16-mixed c2full fp32 c2full bf16-mixed c2 bf16-mixed c5v4 fp32 > gpurun_out/r03m_parity_dist.jsonl 2> gpurun_out/r03m_parity_dist.err; cat gpurun_out/r03m_parity_dist.jsonl | cut -c1-3000; tail -3 gpurun_out/r03m_parity_dist.err
(timeout 300 python -m pytest tests/test_emu_decode.py tests/test_step_parity.py -q -m gpu --timeout 300 -p no:cacheprovider -k "decode or register_staged" 2>&1 | tail -4)
