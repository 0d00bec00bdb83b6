#!/usr/bin/env python3
"""Per-kernel summary (count, total, average, share) of a rocprofv3 --kernel-trace run.

rocprofv3 on ROCm 7.2 writes a rocpd SQLite database (<name>_results.db); this prints the same table
`--stats` would and is what gets committed under profiles/.
Usage: tools/prof_summary.py gpurun_out/prof/bench_results.db [> profiles/<round>_<what>.txt]
"""
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
         "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    dm = demangle([r[0].replace(".kd", "") for r in rows])
    print("# kernels: %d distinct, %d dispatches, %.3f ms total GPU time" % (len(rows), sum(r[1] for r in rows), tot / 1e6))
    print("%-78s %7s %11s %10s %10s %10s %6s %5s %5s %6s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "lds", "scratch"))
    for r in rows:
        name = dm.get(r[0].replace(".kd", ""), r[0])
        name = name if len(name) <= 78 else name[:75] + "..."
        print("%-78s %7d %11.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %6s %7s" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                                          100.0 * r[2] / tot, r[6], r[7], r[9], r[10]))


if __name__ == "__main__":
    main(sys.argv[1])
