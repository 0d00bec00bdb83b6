#!/usr/bin/env python3
"""For every dispatch of kernels matching a pattern in a rocprofv3 rocpd database: the kernels dispatched right before / after it
(counts), to find out which operator a runtime blit or fill belongs to.  tools/prof_neighbours.py <db> <pattern>"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
cnt = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev = rows[i - 1][0][:40] if i else "-"
        nxt = rows[i + 1][0][:40] if i + 1 < len(rows) else "-"
        cnt[(prev, nxt, r[3])] += 1
for k, v in cnt.most_common(40):
    print("%5d  after %-42s before %-42s grid %s" % (v, k[0], k[1], k[2]))
