"""When does the MAIN stream wait for the side stream?  From a rocprofv3 (rocpd SQLite) kernel trace of bench.py: per step (one conv_stem2d_kernel
each) the intervals in which no main-stream kernel runs while a side-stream kernel (the weight gradients) does, summed by the side-stream kernel
running and by where in the step they fall; and the timeline of the step's last kernels.
    python profiles/stream_tail.py /tmp/prof/x_results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols)
rows = list(db.execute(f"select name, start, end, {qcol} from kernels order by start"))
marks = [r[1] for r in rows if "conv_stem2d_kernel" in r[0]]
short = lambda n: n.split("(")[0].replace("void ", "").replace("lp::", "")[:52]  # noqa: E731
main = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
alone = collections.Counter()
where = collections.Counter()
nsteps = 0
for a, b in zip(marks[1:-1], marks[2:]):
    ks = [r for r in rows if a <= r[1] < b]
    nsteps += 1
    pts = sorted(set([a, b] + [r[1] for r in ks] + [min(r[2], b) for r in ks]))
    for t0, t1 in zip(pts, pts[1:]):
        mid = (t0 + t1) / 2
        live = [r for r in ks if r[1] <= mid < r[2]]
        if live and not any(r[3] == main for r in live):
            for r in live:
                alone[short(r[0])] += (t1 - t0) / len(live)
            where[int(10 * (mid - a) / (b - a))] += t1 - t0
print(f"# {nsteps} steps; main stream = {qcol} {main}; ms per step in which only side-stream kernels run: {sum(alone.values()) / nsteps / 1e6:.3f}")
for k, v in alone.most_common(12):
    print(f"  {k:<54} {v / nsteps / 1e6:7.3f}")
print("# ... by tenth of the step:", " ".join(f"{where[i] / nsteps / 1e6:.3f}" for i in range(10)))
a, b = marks[-3], marks[-2]
ks = [r for r in rows if a <= r[1] < b]
print(f"# the last kernels of one step ({(b - a) / 1e6:.3f} ms): start and end in us before the next step's first kernel, stream, name")
for r in ks[-28:]:
    print(f"  {(b - r[1]) / 1e3:9.1f} {(b - r[2]) / 1e3:9.1f}  {'main' if r[3] == main else 'side'}  {short(r[0])}")
