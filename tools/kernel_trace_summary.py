"""Per-kernel durations out of a rocprofv3 --kernel-trace run (the rocpd sqlite database): count, median, mean, p90 per kernel and,
per HIP queue, the gaps between consecutive kernels.  usage: python tools/kernel_trace_summary.py <results.db> [--seq]"""
import collections
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s "
                        "on d.kernel_id = s.id order by d.start"))
by = collections.defaultdict(list)
for st, en, q, name, g, w in rows:
    by[(name.split("(")[0], g // max(w, 1) if "--grid" in sys.argv else 0)].append((en - st) / 1e3)
print(f"{'kernel':90s} {'n':>6s} {'median us':>10s} {'mean us':>10s} {'p90 us':>10s} {'total ms':>10s}")
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{(k + (f' grid {g}' if g else ''))[:90]:90s} {len(v):6d} {v[len(v) // 2]:10.1f} {sum(v) / len(v):10.1f} {v[int(len(v) * .9)]:10.1f} {sum(v) / 1e3:10.2f}")
if "--seq" in sys.argv:
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    for q, rs in byq.items():
        gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
        if gaps:
            print(f"queue {q}: {len(rs)} kernels, gap between consecutive kernels median {statistics.median(gaps):.1f} us, mean {sum(gaps) / len(gaps):.1f} us")
