"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into the per-kernel table `rocprofv3 --stats` prints.

    python profiles/summarize_rocpd.py gpurun_out/prof_r1/r1_results.db > profiles/archive/r01_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}   total kernel time {total / 1e6:.3f} ms   dispatches {sum(r[1] for r in rows)}")
print(f"{'Name':<110} {'Calls':>7} {'TotalDurationNs':>16} {'AverageNs':>12} {'Percentage':>10} {'MinNs':>10} {'MaxNs':>10}")
for name, n, tot, avg, mn, mx in rows:
    print(f"{name[:110]:<110} {n:>7} {tot:>16} {avg:>12.1f} {100 * tot / total:>10.2f} {mn:>10} {mx:>10}")
