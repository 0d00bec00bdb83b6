#!/usr/bin/env python3
"""HBM-side traffic of ONE train step (B=64, N=256), all kernels: per-step FETCH_SIZE x 2 + WRITE_SIZE.

    tools/pmc_train_total.py <out.json>          (on the GPU box; spawns four rocprofv3 passes)

Two workloads of K1 and K2 eager train steps (tools/pmc_workload.py --train-steps K), each under `rocprofv3 --pmc FETCH_SIZE
--kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, MI355X_MICROARCH.md).  Everything that is not a step
(model set-up, the 256 MiB calibration copy, first-step allocations) is in both and cancels:
    bytes per step = (sum over all dispatches of K2 - sum over all dispatches of K1) / (K2 - K1),
per kernel name as well.  FETCH_SIZE counts half of the bytes of wide reads on gfx950 (x2, checked on the copy)."""
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_report  # noqa: E402

K1, K2 = 2, 5
ALGORITHMIC_3X = 3 * 14154240 * 64


def sums(db):
    """kernel name -> (dispatches, total KiB)"""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    q = ("select s.kernel_name, count(*), sum(e.value), max(e.value) from %s e join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id "
         "group by s.kernel_name" % (T("rocpd_pmc_event"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")))
    return {name: (n, tot, mx) for name, n, tot, mx in c.execute(q)}


def main(out_path):
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="rtk_pmc_train_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for k in (K1, K2):
            out = os.path.join(tmp, "%s_%d" % (counter, k))
            subprocess.run([rp, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable,
                            os.path.join(ROOT, "tools", "pmc_workload.py"), "--train-steps", str(k)], cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=600)
            db = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith(".db")][0]
            res[(counter, k)] = sums(db)
    names = set()
    for v in res.values():
        names |= set(v)
    dm = pmc_report.demangle(sorted(names))
    cal = [k for k in res[("WRITE_SIZE", K2)] if "copy" in dm[k].lower()]
    scale, note = 2.0, "no calibration copy found; FETCH_SIZE x 2 per the guide"
    if cal:
        k = max(cal, key=lambda k: res[("WRITE_SIZE", K2)][k][2])
        f_kib = res[("FETCH_SIZE", K2)].get(k, (0, 0, 0))[2]
        if f_kib:
            scale = 262144.0 / f_kib
            note = "256 MiB copy in the same run: FETCH_SIZE max %.0f KiB -> x%.3f; WRITE_SIZE max %.0f KiB (expected 262144)" % (
                f_kib, scale, res[("WRITE_SIZE", K2)][k][2])
    per_kernel = {}
    for name in names:
        g = lambda c, k: res[(c, k)].get(name, (0, 0.0, 0.0))
        calls = (g("FETCH_SIZE", K2)[0] - g("FETCH_SIZE", K1)[0]) / (K2 - K1)
        f = (g("FETCH_SIZE", K2)[1] - g("FETCH_SIZE", K1)[1]) / (K2 - K1) * 1024 * scale
        w = (g("WRITE_SIZE", K2)[1] - g("WRITE_SIZE", K1)[1]) / (K2 - K1) * 1024
        if calls > 0 or f + w > 0:
            per_kernel[dm[name][:90]] = {"launches_per_step": round(calls, 2), "read_bytes": int(f), "written_bytes": int(w), "bytes": int(f + w)}
    total_r = sum(v["read_bytes"] for v in per_kernel.values())
    total_w = sum(v["written_bytes"] for v in per_kernel.values())
    top = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["bytes"]))
    out = {"workload": "one train step (forward + loss + backward + Adam), B=64, N=256, eager",
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, separate passes, %d-step minus %d-step workloads / %d" % (K2, K1, K2 - K1),
           "calibration": note, "read_bytes_per_step": total_r, "written_bytes_per_step": total_w, "bytes_per_step": total_r + total_w,
           "algorithmic_bytes_3x_forward": ALGORITHMIC_3X, "traffic_ratio": round((total_r + total_w) / ALGORITHMIC_3X, 2),
           "kernels_per_step": round(sum(v["launches_per_step"] for v in per_kernel.values()), 1), "by_kernel": top}
    json.dump(out, open(out_path, "w"), indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print("train step: %.2f GB read + %.2f GB written = %.2f GB per step = %.1fx the algorithmic 3 x forward bytes (%s)" % (
        total_r / 1e9, total_w / 1e9, (total_r + total_w) / 1e9, out["traffic_ratio"], note))
    for k, v in list(top.items())[:25]:
        print("  %-90s x%-5.1f %8.1f MB" % (k, v["launches_per_step"], v["bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_pmc_train_total.json"))
