#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc run (rocpd database)."""
import sqlite3, sys, subprocess
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
print("# pmc_event cols:", [r[1] for r in c.execute("pragma table_info(%s)" % pe)])
print("# info_pmc cols:", [r[1] for r in c.execute("pragma table_info(%s)" % pi)])
q = ("select s.kernel_name, p.name, count(*), avg(e.value), min(e.value), max(e.value) from %s e join %s p on e.pmc_id=p.id "
     "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, p.name order by 4 desc" % (pe, pi, kd, ks))
for r in c.execute(q):
    name = subprocess.run(["c++filt", r[0].replace(".kd", "")], capture_output=True, text=True).stdout.strip()[:80]
    print("%-82s %-12s n=%4d avg=%14.1f min=%14.1f max=%14.1f" % (name, r[1], r[2], r[3], r[4], r[5]))
