#!/usr/bin/env python3
"""What bounds sa_scale_split_kernel?  (round 5)  Patched COPIES of csrc/fused_split.hip (the product source carries no ablation
macros), each linked into ratrack_amd/lib/variants/librtk_sa_<tag>.so, all timed in one GPU call on the largest scale of the bench
batch (128 clouds x 256 live centroids x 32 neighbours, 64 -> 64 channels).  Results of the ablations are WRONG by construction;
only their times mean anything.

    python tools/experiments/sa_ablate.py --build     (CPU: hipcc)
    python tools/experiments/sa_ablate.py             (GPU: runs every variant in a subprocess)
"""
import argparse
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")

VARIANTS = {
    "base": [],
    # every position gathers ROW 0 of its sample: is the kernel bound by the spread of the gather?
    "row0": [("const float *qrow = P.q + ((long)b * P.n + (id < src_e ? id : 0)) * P.q_pitch + 4 * hh;",
              "const float *qrow = P.q + ((long)b * P.n + (id < 0 ? id : 0)) * P.q_pitch + 4 * hh;")],
    # no gather at all: the lane's coordinates stand in for the row
    "nogather": [("                const f4 t = *reinterpret_cast<const f4 *>(qrow + 32 * v + 8 * q);\n                a[4 * q] = t.x;",
                  "                const f4 t = (f4){dx, dy, dz, (float)(v + q)}; (void)qrow;\n                a[4 * q] = t.x;")],
    # no split layer: the accumulators take the activations
    "nolayer2": [("            RTK_SA_MM(1, 0) RTK_SA_MM(0, 1) RTK_SA_MM(0, 0)\n", "            acc[0][s] += __uint_as_float(bp[0][0] ^ fr[0][0][0]); acc[1][s] += __uint_as_float(bp[1][1] ^ fr[1][1][1]);\n")],
    # ball indices and coordinates of a fixed pattern instead of loaded (removes the idx -> xyz / row dependent chain)
    "noidx": [("        const int id = P.idx[(long)c * NS + slot];", "        const int id = (c * 7 + slot * 13) & 255;")],
}


def build():
    from ratrack_amd import build as B
    B.build(verbose=False)
    src0 = open(os.path.join(B.CSRC, "fused_split.hip")).read()
    os.makedirs(VAR, exist_ok=True)
    for tag, patches in VARIANTS.items():
        src = src0
        for a, r in patches:
            assert src.count(a) == 1, (tag, a)
            src = src.replace(a, r)
        patched = os.path.join(VAR, "fused_split_sa_%s.hip" % tag)
        open(patched, "w").write(src)
        obj = patched[:-4] + ".o"
        subprocess.check_call([B._hipcc()] + B.flags_for("fused_split.hip") + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", patched, "-o", obj])
        objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if os.path.basename(o) != "fused_split.o"]
        out = os.path.join(VAR, "librtk_sa_%s.so" % tag)
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", out] + objs + [obj])
        os.remove(obj)
        os.remove(patched)
        print(out)


def run_one(so):
    import ratrack_amd._lib as L
    L.SO_PATH = so
    import torch
    from ratrack_amd import fused as F
    dev = "cuda"
    torch.manual_seed(0)
    for (ns, c1, n, npoint, live) in ((32, 64, 512, 512, 256), (16, 64, 512, 512, 256), (16, 32, 512, 512, 256)):
        samples = 128
        xyz, new_xyz = torch.randn(samples, n, 3, device=dev), torch.randn(samples, npoint, 3, device=dev)
        idx = torch.randint(0, live, (samples, npoint, ns), device=dev, dtype=torch.int32)
        q = torch.randn(samples * n, c1, device=dev)
        w1img = F.offset_image((torch.randn(c1, 4, device=dev) * 0.3).double(), dev)
        w2, b2 = torch.randn(64, c1, device=dev) / 8, torch.randn(64, device=dev) * 0.1
        img, inv = F.pack_split_device(w2)
        nu = torch.full((samples,), live, device=dev, dtype=torch.int32)
        out = torch.zeros(samples * npoint, 128, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def launch():
            L.call("rtk_sa_scale_split", samples, n, npoint, ns, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), q.data_ptr(), c1, c1,
                   w1img.data_ptr(), img.data_ptr(), inv.data_ptr(), b2.data_ptr(), out.data_ptr(), 128, 0, nu.data_ptr(), nu.data_ptr(), st)
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        pos = samples * live * ns
        print("   ns=%2d c1=%2d: %7.1f us   (%.2f TB/s of gathered rows, %.1f TFLOP/s of the %d -> 64 layer)"
              % (ns, c1, us, pos * c1 * 4 / us / 1e6, 2.0 * pos * c1 * 64 / us / 1e6, c1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--one", default=None)
    a = ap.parse_args()
    if a.build:
        return build()
    if a.one:
        return run_one(a.one)
    for tag in VARIANTS:
        so = os.path.join(VAR, "librtk_sa_%s.so" % tag)
        print("%s:" % tag, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", so], check=False)


if __name__ == "__main__":
    main()
