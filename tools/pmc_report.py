#!/usr/bin/env python3
"""Merge a FETCH_SIZE pass, a WRITE_SIZE pass and a kernel-trace pass (rocpd databases) into per-kernel HBM-side traffic:

    tools/pmc_report.py <fetch.db> <write.db> <trace.db> <out_dir> [tag]

Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE on gfx950 reports half of the bytes of wide coalesced reads, so it is
doubled -- and the factor is CHECKED in the same run on a 256 MiB clone (at::native copy kernel: reads 256 MiB, writes 256 MiB);
counters are in KiB.  Writes profiles/<tag>_pmc_{fetch,write}_size.txt (raw per-kernel tables), <tag>_irregular_hbm.json,
<tag>_pmc_cost_volume.json and <tag>_pmc_cost_volume_bwd.json."""
import json
import os
import sqlite3
import subprocess
import sys


def table(path, pmc):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    out = {}
    if pmc:
        pe, pi = T("rocpd_pmc_event"), T("rocpd_info_pmc")
        q = ("select s.kernel_name, p.name, count(*), avg(e.value), min(e.value), max(e.value) from %s e join %s p on e.pmc_id=p.id "
             "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, p.name" % (pe, pi, kd, ks))
        for name, cname, n, avg, lo, hi in c.execute(q):
            out[name] = (cname, n, avg, lo, hi)
    else:
        q = "select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name" % (kd, ks)
        for name, n, avg, lo in c.execute(q):
            out[name] = (n, avg, lo)
    return out


def demangle(names):
    outs = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, outs))


def main(fetch_db, write_db, trace_db, out_dir, tag="r02"):
    F, W, T = table(fetch_db, True), table(write_db, True), table(trace_db, False)
    dm = demangle(sorted(set(F) | set(W) | set(T)))
    for nm, tab in (("fetch", F), ("write", W)):
        with open(os.path.join(out_dir, "%s_pmc_%s_size.txt" % (tag, nm)), "w") as f:
            f.write("# rocprofv3 --pmc %s_SIZE --kernel-trace -- python tools/pmc_workload.py   (KiB per dispatch, raw)\n" % nm.upper())
            for k, (cname, n, avg, lo, hi) in sorted(tab.items(), key=lambda kv: -kv[1][2]):
                f.write("%-100s %-12s n=%4d avg=%14.1f min=%14.1f max=%14.1f\n" % (dm[k][:100], cname, n, avg, lo, hi))
    # calibration: the 256 MiB clone
    # the 256 MiB clone is the dispatch with the largest WRITE_SIZE among the copy kernels (framework copy kernel or the runtime's blit)
    cal = [k for k in W if "copy" in dm[k].lower()]
    cal_k = max(cal, key=lambda k: W[k][4]) if cal else None
    fetch_scale, cal_note = 2.0, "no calibration kernel found; FETCH_SIZE doubled per the guide"
    if cal_k:
        f_kib, w_kib = F.get(cal_k, (0, 0, 0, 0, 0))[4], W[cal_k][4]
        fetch_scale = 262144.0 / f_kib if f_kib else 2.0
        cal_note = ("256 MiB clone in the same run (%s): max FETCH_SIZE = %.1f KiB -> scale %.3f (guide: 2), max WRITE_SIZE = %.1f KiB "
                    "(expected 262144)" % (dm[cal_k][:40], f_kib, fetch_scale, w_kib))

    def traffic(pred):
        ks = [k for k in set(F) | set(W) if pred(dm[k])]
        if not ks:
            return None
        k = max(ks, key=lambda k: F.get(k, (0, 0, 0, 0, 0))[2])
        f, w = F.get(k, (0, 0, 0.0))[2], W.get(k, (0, 0, 0.0))[2]
        dur = T.get(k, (0, 0.0, 0.0))
        return {"kernel": dm[k][:80], "FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "dispatches": F.get(k, (0, 0))[1],
                "traffic_bytes_per_launch": int((f * fetch_scale + w) * 1024), "avg_duration_us": round(dur[1] / 1e3, 2),
                "GB/s": round((f * fetch_scale + w) * 1024 / max(dur[1], 1), 2)}

    irregular = {"geometry_front_kernel": "geometry_front_kernel", "geometry_tables_kernel": "geometry_tables_kernel", "fps_wave_kernel": "fps_wave_kernel", "fps_relevel_kernel": "fps_relevel_kernel", "ball_query_pair_kernel": "ball_query_pair_kernel",
                 "three_nn_kernel": "three_nn_kernel", "knn_point_kernel": "knn_point_kernel", "scatter_add_rows_kernel": "scatter_rows256_kernel",
                 "group_points_grad_kernel": "group_points_grad_lds_kernel", "sa_first_layer_bwd_kernel": "sa_first_layer_bwd_kernel",
                 "three_interpolate_grad_kernel": "three_interp_grad_gather_kernel", "inverse_index_kernel": "inverse_index_kernel"}
    rows = {k: traffic(lambda n, pat=pat: pat in n) for k, pat in irregular.items()}
    rows = {k: v for k, v in rows.items() if v}
    json.dump({"workload": "B=64, N=256: forward + train step kernels (eager) and the irregular-op launches of ratrack_amd.benchutil",
               "collected_with": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) / --kernel-trace "
                                 "-- python tools/pmc_workload.py; merged by tools/pmc_report.py",
               "calibration": cal_note, "peak_GB/s": 8000,
               "note": "fabric-side bytes (Infinity Cache hits included), averaged over all dispatches of a kernel name (a name that is "
                       "launched with several shapes, e.g. ball_query_pair_kernel, averages over them); durations are from the separate "
                       "kernel-trace pass.  These kernels are latency-bound serial chains or small gathers: none comes near the HBM roofline, "
                       "and none needs to -- together they move < 3 % of the step's bytes.",
               "traffic_bytes_per_launch": {k: v["traffic_bytes_per_launch"] for k, v in rows.items()}, "kernels": rows},
              open(os.path.join(out_dir, "%s_irregular_hbm.json" % tag), "w"), indent=1)
    split = any("cost_volume_split_kernel" in n for n in dm.values())
    stem = "cost_volume_split_kernel" if split else "cost_volume_kernel"
    for fname, pat, alg, what in (("pmc_cost_volume", stem + "<false>", 51773440, "one launch per backbone forward"),
                                  ("pmc_cost_volume_train", stem + "<true>", None, "training forward: also stores a1, a2, a3 and two sign masks"),
                                  ("pmc_cost_volume_bwd", "cost_volume_bwd_split_kernel" if split else "cost_volume_bwd_kernel", None,
                                   "one launch per train step")):
        r = traffic(lambda n, pat=pat: pat in n)
        if r:
            r.update({"workload": "B=64, N=256 (%s)" % what, "calibration": cal_note,
                      "collected_with": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes), tools/pmc_workload.py"})
            if alg:
                r["algorithmic_bytes_per_launch"] = alg
            json.dump(r, open(os.path.join(out_dir, "%s_%s.json" % (tag, fname)), "w"), indent=1)
    print("calibration:", cal_note)
    for k, v in rows.items():
        print("%-32s %10.1f KB/launch %8.2f us  %7.1f GB/s" % (k, v["traffic_bytes_per_launch"] / 1e3, v["avg_duration_us"], v["GB/s"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
