"""MFMA utilisation and issue-stall picture of the top kernels from one rocprofv3 PMC pass of SQ counters.

    python profiles/summarize_pmc_sq.py <sq_results.db> > profiles/archive/r02_pmc_mfma.json

Units (/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots" and the s_memtime table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(32 per v_mfma_f32_32x32x16_bf16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_BUSY_CYCLES counts cycles the
SQ had any wave, summed over the shader engines it is replicated in.  mfma_util = MFMA_BUSY / (4 SIMDs x busy cycles of the CUs): reported
here as MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES-equivalent when available, else against SQ_BUSY_CYCLES (then a relative figure between
kernels, not an absolute one); the wave-level split WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY is a fraction of SQ_WAVE_CYCLES."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
by: dict = {}
for k, c, n, v in rows:
    name = k.split("(")[0].replace("void ", "")
    d = by.setdefault(name, {"launches": n})
    d[c] = v
dur = {k.split("(")[0].replace("void ", ""): (n, t) for k, n, t in
       db.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()} if "kernels" in [r[0] for r in db.execute(
           "select name from sqlite_master").fetchall()] else {}
out = {}
for name, d in by.items():
    if "lp::" not in name:
        continue
    wave = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    busy = d.get("SQ_BUSY_CYCLES", 0.0) or 1.0
    mfma = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rec = {"launches": d["launches"], "mfma_busy_cycles": mfma, "sq_busy_cycles": busy, "mfma_busy_per_sq_busy": round(mfma / busy, 4),
           "mfma_mops_bf16": d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
           "wave_cycles_split": {"wait_any": round(d.get("SQ_WAIT_ANY", 0.0) / wave, 4), "wait_inst_any": round(d.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
                                 "active_inst_any": round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4)},
           "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT")}
    if name in dur:
        rec["total_ms"] = round(dur[name][1] / 1e6, 3)
        # absolute utilisation: busy cycles summed over the chip's 256 CUs x 4 SIMDs, against the kernels' wall time at the 2.4 GHz engine clock
        rec["mfma_util_vs_2.4GHz_x_1024_simd"] = round(mfma / (1024.0 * dur[name][1] * 2.4), 4)
    out[name] = rec
top = dict(sorted(out.items(), key=lambda kv: -kv[1].get("total_ms", kv[1]["mfma_busy_cycles"]))[:12])
print(json.dumps({"source": sys.argv[1], "kernels": top}, indent=1))
