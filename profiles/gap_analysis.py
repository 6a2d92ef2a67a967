"""Where a step's wall time goes, from a rocprofv3 (rocpd SQLite) kernel trace: per step (one `conv_stem2d_kernel` each) the time some kernel runs,
the time two or more run (side-stream weight gradients under the main stream), the idle time, and the idle gaps grouped by the kernels either side.

    python profiles/gap_analysis.py /tmp/prof/x_results.db > profiles/archive/r03_gap_analysis.txt
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns of `kernels`:", ", ".join(cols))
qcol = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
rows = list(db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
marks = [r[1] for r in rows if "conv_stem2d_kernel" in r[0] or "conv_igemm_kernel<64, 2>" in r[0]]
if len(marks) < 3:
    raise SystemExit("no step markers")
short = lambda n: n.split("(")[0].replace("void ", "").replace("lp::", "")[:48]  # noqa: E731
tot = collections.Counter()
gaps = collections.Counter()
gapn = collections.Counter()
nsteps = 0
for a, b in zip(marks[1:-1], marks[2:]):   # skip the first (warm-up) step
    ks = [r for r in rows if a <= r[1] < b]
    nsteps += 1
    ev = sorted([(r[1], 1, i) for i, r in enumerate(ks)] + [(min(r[2], b), -1, i) for i, r in enumerate(ks)])
    live, t0, last_end_name = 0, a, None
    for t, d, i in ev:
        dt = t - t0
        if dt > 0:
            tot["idle" if live == 0 else "one kernel" if live == 1 else "two or more"] += dt
            if live == 0 and d == 1:
                key = (last_end_name, short(ks[i][0]))
                gaps[key] += dt
                gapn[key] += 1
        live += d
        if d == -1:
            last_end_name = short(ks[i][0])
        t0 = t
    tot["wall"] += b - a
    tot["sum of kernel durations"] += sum(r[2] - r[1] for r in ks)
    if qcol:
        for q in set(r[3] for r in ks):
            tot[f"busy, {qcol} {q}"] += sum(r[2] - r[1] for r in ks if r[3] == q)
print(f"# {nsteps} steps; ms per step")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:<32} {v / nsteps / 1e6:8.3f}")
print("# idle gaps by (kernel that ended, kernel that started): ms per step, gaps per step, mean us")
for key, v in gaps.most_common(40):
    print(f"{str(key[0]):<50} -> {key[1]:<50} {v / nsteps / 1e6:7.3f} {gapn[key] / nsteps:6.1f} {v / gapn[key] / 1e3:7.1f}")
