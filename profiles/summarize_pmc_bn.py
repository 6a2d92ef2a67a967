"""HBM traffic and rate of the BatchNorm streaming kernels from the same two rocprofv3 PMC passes summarize_pmc.py reads (FETCH_SIZE,
WRITE_SIZE in KiB, reads doubled as the microarchitecture guide prescribes for 16-B/lane streams on gfx950) + the kernel durations of the
FETCH pass's own kernel trace:

    python profiles/summarize_pmc_bn.py <fetch.db> <write.db> > profiles/r04_pmc_bn_traffic.json
"""
import json
import sqlite3
import sys

NAMES = ("bn_apply_kernel", "bn_bwd_apply_kernel", "colreduce_kernel", "bn_relu_maxpool_fwd_kernel", "bn_pool_bwd_v2_kernel")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? "
                     "group by kernel_name", (counter,)).fetchall()
    return {k.split('(')[0].replace('void ', '').replace('lp::', ''): (n, v) for k, n, v in rows if any(s in k for s in NAMES)}


def durations(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master").fetchall()]
    if "kernels" not in tabs:
        return {}
    rows = c.execute("select name, count(*), sum(end - start) from kernels group by name").fetchall()
    return {k.split('(')[0].replace('void ', '').replace('lp::', ''): (n, t) for k, n, t in rows if any(s in k for s in NAMES)}


fetch, write, dur = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), durations(sys.argv[1])
out = {}
for k in sorted(fetch):
    n, f = fetch[k]
    w = write.get(k, (n, 0.0))[1]
    hbm = (2.0 * f + w) * 1024.0
    rec = {"launches": n, "hbm_read_bytes_per_launch": 2.0 * f * 1024.0 / n, "hbm_write_bytes_per_launch": w * 1024.0 / n}
    if k in dur and dur[k][1] > 0:
        rec["avg_us_under_the_counters"] = dur[k][1] / dur[k][0] / 1e3
        rec["hbm_gbs"] = hbm / dur[k][1]          # bytes per ns = GB/s
    out[k] = rec
print(json.dumps({"correction": "reads = 2 x FETCH_SIZE (gfx950 16-B/lane streams), writes = WRITE_SIZE", "by_kernel": out}, indent=1))
