#!/bin/bash
# r06f: decode tests on the device (the new strip accumulation incl. the three-wave case), then ONE counter pass over the whole step that asks every kernel the
# question that found the decode's LDS atomics: how busy is the LDS, how many wave cycles wait for it (SQ_LDS_IDX_ACTIVE, SQ_WAIT_INST_LDS per kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_emu_decode.py tests/test_step_parity.py -q -m gpu -x -p no:cacheprovider -k "decode or repeats or c2full" 2>&1 | tail -3) | tee gpurun_out/r06f_pytest.txt
CMD="python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d /tmp/r06f_pmc -o lds -- $CMD > gpurun_out/r06f_pmc.log 2>&1
python profiles/summarize_pmc_any.py /tmp/r06f_pmc/lds_results.db lp:: > gpurun_out/r06f_lds_by_kernel.json 2>> gpurun_out/r06f_pmc.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06f_lds_by_kernel.json"))
rows=[]
for k,v in d.items():
    if "total_us" not in v or not v.get("SQ_WAVE_CYCLES"): continue
    cyc=v["total_us"]*1e-6*2.1e9*256      # CU-cycles of the launches (2.1 GHz under load)
    rows.append((v["total_us"], k[:70], v.get("SQ_LDS_IDX_ACTIVE",0)/cyc, v.get("SQ_WAIT_INST_LDS",0)/v["SQ_WAVE_CYCLES"], v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1), v.get("SQ_ACTIVE_INST_VALU",0)/v["SQ_WAVE_CYCLES"]))
rows.sort(reverse=True)
print("total_us  LDS-busy  waitLDS/wave  conflict/LDSbusy  VALU/wave  kernel")
for r in rows[:40]: print("%9.0f  %7.2f  %11.2f  %15.2f  %9.2f  %s"%(r[0],r[2],r[3],r[4],r[5],r[1]))
PY
