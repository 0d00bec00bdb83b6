#!/usr/bin/env python3
"""Dispatch order of the last full repetition in a rocprofv3 rocpd database: the kernels between the last two dispatches of a
marker kernel, run-length compressed, with start offsets (us).  tools/prof_sequence.py <db> <marker pattern> [max rows] [interval, 1 = last]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
marks = [i for i, r in enumerate(rows) if pat in r[0]]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
a, b = marks[-k - 1], marks[-k]
t0 = rows[a][1]; out = []; run = None
for r in rows[a:b + 1]:
    key = (r[0][:70], r[3])
    if run and run[0] == key:
        run[1] += 1; run[3] = (r[2] - t0) / 1e3
    else:
        run = [key, 1, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3]; out.append(run)
for key, n, s, e in out[:lim]:
    print("%9.1f %9.1f  x%-3d grid %-8d %s" % (s, e, n, key[1], key[0]))
seg = rows[a:b + 1]
busy = sum(r[2] - r[1] for r in seg) / 1e3
gaps = sorted(((seg[i + 1][1] - max(x[2] for x in seg[max(0, i - 3):i + 1])) / 1e3, seg[i][0][:50], seg[i + 1][0][:50]) for i in range(len(seg) - 1))
print("span %.1f us, sum of kernel durations %.1f us, %d dispatches; idle gaps > 1 us: %.1f us" % (
    (seg[-1][2] - seg[0][1]) / 1e3, busy, len(seg), sum(g[0] for g in gaps if g[0] > 1)))
for g in gaps[::-1][:12]:
    print("  gap %7.1f us  after %-50s before %s" % g)
