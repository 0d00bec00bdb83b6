#!/usr/bin/env python3
"""Round 5: A/B of a library variant in the pipelined forward, one box, alternating runs.
    python tools/experiments/ab_lib.py --build ops_pointnet2.hip -DFPS_WPB=1 --tag fps1     (CPU: variant .so under lib/variants)
    python tools/experiments/ab_lib.py --run fps1 [--reps 3] [--bench-args ...]              (GPU)"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")


def build(src, defs, tag):
    from ratrack_amd import build as B
    B.build(verbose=False)
    os.makedirs(VAR, exist_ok=True)
    obj = os.path.join(VAR, "ab_%s.o" % tag)
    subprocess.check_call([B._hipcc()] + B.flags_for(src) + defs + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", os.path.join(B.CSRC, src), "-o", obj])
    objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if os.path.basename(o) != src.replace(".hip", ".o")]
    out = os.path.join(VAR, "librtk_ab_%s.so" % tag)
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", out] + objs + [obj])
    os.remove(obj)
    print(out)


def one(so, bench_args):
    code = ("import sys, io, json, contextlib, runpy\n"
            "import ratrack_amd._lib as L\n"
            + ("L.SO_PATH = %r\n" % so if so else "") +
            "sys.argv = ['bench.py'] + %r\n" % bench_args +
            "buf = io.StringIO()\n"
            "with contextlib.redirect_stdout(buf):\n"
            "    try:\n        runpy.run_path('bench.py', run_name='__main__')\n    except SystemExit:\n        pass\n"
            "d = json.loads(buf.getvalue().strip().splitlines()[-1])\n"
            "print(json.dumps({'value': d['value'], 'ms': d['ms_per_step'], 'cv_insitu_ms': d.get('roofline', {}).get('kernel_ms')}))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", default=None)
    ap.add_argument("--tag", default="variant")
    ap.add_argument("--run", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bench-args", default="--no-cpu-baseline --no-train --no-configs --traffic off --seconds 3")
    a, rest = ap.parse_known_args()
    if a.build:
        return build(a.build, rest, a.tag)
    so = os.path.join(VAR, "librtk_ab_%s.so" % a.run)
    for _ in range(a.reps):
        print("tree   ", one(None, a.bench_args.split()), flush=True)
        print("%-7s" % a.run, one(so, a.bench_args.split()), flush=True)


if __name__ == "__main__":
    main()
