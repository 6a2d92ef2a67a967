"""Per-kernel sums of whatever counters one rocprofv3 --pmc pass collected (+ kernel durations):  python profiles/summarize_pmc_any.py <results.db> [filter]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "lp::"
tabs = [r[0] for r in db.execute("select name from sqlite_master").fetchall()]
rows = db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
out = {}
for k, c, n, v in rows:
    if flt not in k:
        continue
    name = k.split("(")[0].replace("void ", "")
    d = out.setdefault(name, {"launches": n})
    d[c] = v
if "kernels" in tabs:
    for k, n, t in db.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall():
        name = k.split("(")[0].replace("void ", "")
        if name in out:
            out[name]["total_us"] = round(t / 1e3, 1)
print(json.dumps(out, indent=1))
