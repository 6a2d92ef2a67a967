"""HBM traffic of the MFMA convolution kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters in KiB).

    python profiles/summarize_pmc.py <fetch.db> <write.db> > profiles/archive/r01_pmc_traffic.json

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies a wide coalesced 16 B/lane stream
at HALF its bytes, so reads are doubled; WRITE_SIZE matched the known output bytes of the conv micro-benchmark exactly
(profiles/conv_microbench.py: 604 MB written -> 589 824 KiB reported).
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("conv.hip", "conv_pipe.h", "conv_res2d.h", "conv_stem_wgrad.h", "lp_common.h")   # what the measured kernels are compiled from (bench.py checks the same digest)


def kernels_sha256():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "lightning-pose_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()



def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? "
                     "and (kernel_name like '%lp::conv%') group by kernel_name", (counter,)).fetchall()
    return {k.split('(')[0].replace('void lp::', ''): (n, v) for k, n, v in rows}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out, tot_b, tot_n = {}, 0.0, 0
for k in sorted(fetch):
    n, f = fetch[k]
    w = write.get(k, (n, 0.0))[1]
    hbm = (2.0 * f + w) * 1024.0
    out[k] = {"launches": n, "fetch_kib_reported": f, "write_kib_reported": w, "hbm_bytes_per_launch": hbm / n}
    tot_b += hbm
    tot_n += n
print(json.dumps({"conv_hbm_bytes_per_launch": tot_b / max(tot_n, 1), "conv_launches": tot_n, "kernels_sha256": kernels_sha256(),
                  "correction": "reads = 2 x FETCH_SIZE (gfx950 16-B/lane streams), writes = WRITE_SIZE", "by_kernel": out}, indent=1))
