"""Every kernel of a rocprofv3 --kernel-trace run (rocpd sqlite database) as one line: start and end in us since the first, queue, grid
in workgroups, name -- for reading the device time line of a one-shot call.  usage: python tools/kernel_timeline.py <results.db> > out.tsv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s "
                        "on d.kernel_id = s.id order by d.start"))
t0 = rows[0][0] if rows else 0
for st, en, q, name, g, w in rows:
    short = name.split("(")[0].replace("void phmm::", "").replace("phmm::", "")
    print(f"{(st - t0) / 1e3:12.1f}\t{(en - t0) / 1e3:12.1f}\t{(en - st) / 1e3:9.1f}\tq{q}\t{g // max(w, 1):7d}\t{short}")
