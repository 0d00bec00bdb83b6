#!/usr/bin/env python3
"""Where does a tile of cost_volume_split_kernel spend its time?  (round 4)

Builds a copy of ratrack_amd/csrc/fused_split.hip with clock reads at the phase boundaries of the tile loop (the product source has
no instrumentation), links it with the cached objects of the regular build into ratrack_amd/lib/variants/librtk_cvticks.so, runs the
forward cost volume alone at the bench shape and prints, per phase, the mean over all waves of the clock ticks per tile.

    python tools/experiments/cv_ticks.py --build        (CPU: hipcc only)
    python tools/experiments/cv_ticks.py [--batch 64]   (GPU)
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
SO = os.path.join(VAR, "librtk_cvticks.so")
NT = 16

HEAD = r'''
__device__ unsigned long long g_cv_ticks[1024 * 16];
#define CV_TICK(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); tk[k] += t_ - tprev; tprev = t_; \
                     __builtin_amdgcn_sched_barrier(0); }
'''
TAIL = r'''
extern "C" __attribute__((visibility("default"))) int rtk_dbg_cv_ticks(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cv_ticks), sizeof(g_cv_ticks));
}
'''
PATCHES = [
    # (anchor, replacement) -- each anchor must occur exactly once in the forward kernel
    ("    for (int t = t0; t < ntiles; t += tstep) {\n        asm volatile(\"\" ::: \"memory\");\n        // the next tile's neighbour index",
     "    unsigned long long tk[%d] = {}, tprev = __builtin_readcyclecounter(), t00 = tprev, w00 = wall_clock64();\n"
     "    for (int t = t0; t < ntiles; t += tstep) {\n        asm volatile(\"\" ::: \"memory\");\n        CV_TICK(0)\n        // the next tile's neighbour index" % NT),
    ("        // byte offset of this lane's first 16-byte slot in a (position, 256) row (one 32-bit VGPR on uniform base pointers)\n        const long pos = i * 16 + j;",
     "        CV_TICK(1)\n        const long pos = i * 16 + j;"),
    ("        LaneScale sc = lane_scale32(h);\n        split_layer<0>(ws, h, sc.s, acc, SidePair<CvLayer2Side<SAVE>",
     "        LaneScale sc = lane_scale32(h);\n        CV_TICK(12)\n        split_layer<0>(ws, h, sc.s, acc, SidePair<CvLayer2Side<SAVE>"),
    ("        float c = sc.inv * wi2;\n", "        CV_TICK(2)\n        float c = sc.inv * wi2;\n"),
    ("        sc = lane_scale32(h);\n        if (SAVE) split_layer<SPLIT_NF>", "        sc = lane_scale32(h);\n        CV_TICK(3)\n        if (SAVE) split_layer<SPLIT_NF>"),
    ("        ws.sync();                                                   // wrap the stream to chunk 0\n        c = sc.inv * wi3;\n",
     "        CV_TICK(4)\n        ws.sync();                                                   // wrap the stream to chunk 0\n        CV_TICK(5)\n        c = sc.inv * wi3;\n"),
    ("        WnBlock wk = wn_block(P.wn, 0, hh, col);                     // block 0's operands travel during the hidden layers",
     "        CV_TICK(13)\n        WnBlock wk = wn_block(P.wn, 0, hh, col);                     // block 0's operands travel during the hidden layers"),
    ("        float *o = P.out + i * P.out_pitch + 4 * hh;\n        f16v wpre", "        CV_TICK(6)\n        float *o = P.out + i * P.out_pitch + 4 * hh;\n        f16v wpre"),
    ("        out_block(0);\n        out_block(1);\n", "        out_block(0);\n        CV_TICK(7)\n        out_block(1);\n        CV_TICK(8)\n"),
    ("#pragma unroll\n        for (int v = 2; v < SPLIT_VB; ++v) out_block(v);\n        pt = ptn;",
     "        CV_TICK(9)\n#pragma unroll\n        for (int v = 2; v < SPLIT_VB; ++v) out_block(v);\n        CV_TICK(10)\n        pt = ptn;"),
    ("        dx = __fsub_rn(cn[0], cn[3]); dy = __fsub_rn(cn[1], cn[4]); dz = __fsub_rn(cn[2], cn[5]);\n    }\n    ws.finish();\n}",
     "        dx = __fsub_rn(cn[0], cn[3]); dy = __fsub_rn(cn[1], cn[4]); dz = __fsub_rn(cn[2], cn[5]);\n        CV_TICK(11)\n    }\n    ws.finish();\n}"),
]
FINAL = ("    ws.finish();\n}\n\n// ---- rtk_cost_volume_bwd on the split path",
         "    ws.finish();\n    tk[14] = __builtin_readcyclecounter() - t00; tk[15] = wall_clock64() - w00;\n"
         "    if (lane == 0) for (int k = 0; k < %d; ++k) g_cv_ticks[(blockIdx.x * SP_NW + wave) * 16 + k] = tk[k];\n}\n\n"
         "// ---- rtk_cost_volume_bwd on the split path" % NT)
BWD_PATCHES = [
    ("    for (int G = bx; G < groups; G += nbx) {\n        asm volatile(\"\" ::: \"memory\");\n        const long pos = i * 16 + j;\n        const unsigned ro = (unsigned)pos * 1024u + 16u * hh;\n        if (valid && hh == 0) *reinterpret_cast<f4 *>(Q.d4",
     "    unsigned long long tk[%d] = {}, tprev = __builtin_readcyclecounter(), t00 = tprev, w00 = wall_clock64();\n"
     "    for (int G = bx; G < groups; G += nbx) {\n        asm volatile(\"\" ::: \"memory\");\n        CV_TICK(0)\n        const long pos = i * 16 + j;\n        const unsigned ro = (unsigned)pos * 1024u + 16u * hh;\n        if (valid && hh == 0) *reinterpret_cast<f4 *>(Q.d4" % NT),
    ("        // ---- da2 = W3^T dz3;", "        CV_TICK(1)\n        // ---- da2 = W3^T dz3;"),
    ("        split_layer<0>(ws, h, acc, StoreRowsSide{Q.dz3, ro, valid});\n", "        CV_TICK(2)\n        split_layer<0>(ws, h, acc, StoreRowsSide{Q.dz3, ro, valid});\n        CV_TICK(3)\n"),
    ("        // ---- da1 = W2^T dz2;", "        CV_TICK(4)\n        // ---- da1 = W2^T dz2;"),
    ("        split_layer<SPLIT_NF>(ws, h, acc, StoreRowsSide{Q.dz2, ro, valid});\n        ws.sync();",
     "        CV_TICK(5)\n        split_layer<SPLIT_NF>(ws, h, acc, StoreRowsSide{Q.dz2, ro, valid});\n"
     "        CV_TICK(6)\n        ws.sync();\n        CV_TICK(7)"),
    ("        pt = ptn; valid = validn; i = in_; dx = dxn; dy = dyn; dz = dzn;\n    }\n    ws.finish();",
     "        CV_TICK(8)\n        pt = ptn; valid = validn; i = in_; dx = dxn; dy = dyn; dz = dzn;\n    }\n    ws.finish();"),
]
BWD_FINAL = ("    ws.finish();\n}\n\n// ---- rtk_sa_scale on the split path",
             "    ws.finish();\n    tk[14] = __builtin_readcyclecounter() - t00; tk[15] = wall_clock64() - w00;\n"
             "    if (lane == 0) for (int k = 0; k < %d; ++k) g_cv_ticks[(blockIdx.x * SP_NW + wave) * 16 + k] = tk[k];\n}\n\n"
             "// ---- rtk_sa_scale on the split path" % NT)
BWD_NAMES = ["loop top", "masks, a3 rows, WeightNet, dz3 / dq3 / dt2 (128 K = 2 MFMAs), dq3 stores", "zero accumulators", "layer W3^T (64 group steps, dz3 stores ride along)",
             "dz2 = da2 leaky'(z2), bias row sums", "zero accumulators", "layer W2^T (64 group steps, dz2 stores)", "stream wrap sync",
             "epilogue: dz1 stores, dp1 / dWd neighbour sums, next tile's index and direction"]
SAME_ROW = ("*r2 = P.p2 + nb * 256 + 4 * hh;", "*r2 = P.p2 + (long)b * P.n2 * 256 + 4 * hh;")   # --same-row: every lane gathers row 0 of its sample (wrong results)


def build(same_row=False, src_path=None, totals_only=False, out=SO, inc=(), defs=(), bwd=False):
    from ratrack_amd import build as B
    B.build(verbose=False)
    src = open(src_path or os.path.join(B.CSRC, "fused_split.hip")).read()
    patches, final = (BWD_PATCHES, BWD_FINAL) if bwd else (PATCHES, FINAL)
    for anchor, repl in (patches[:1] if totals_only else patches) + [final] + ([SAME_ROW] if same_row else []):
        assert src.count(anchor) == 1, "anchor not unique / not found:\n" + anchor
        src = src.replace(anchor, repl)
    k = src.index("namespace {") if "namespace {" in src[:3000] else src.index("#include \"split_mfma.h\"") + len("#include \"split_mfma.h\"")
    k = src.index("\n", src.index("#include \"split_mfma.h\"")) + 1
    src = src[:k] + HEAD + src[k:] + TAIL
    os.makedirs(VAR, exist_ok=True)
    patched = os.path.join(VAR, "fused_split_cvticks.hip")
    open(patched, "w").write(src)
    obj = patched[:-4] + ".o"
    subprocess.check_call([B._hipcc()] + B.flags_for("fused_split.hip") + ["-D" + d for d in defs] + ["-I", os.path.join(ROOT, "include")] + [x for d in inc for x in ("-I", d)] + ["-I", B.CSRC, "-c", patched, "-o", obj])
    objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if os.path.basename(o) != "fused_split.o"]
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", out] + objs + [obj])
    os.remove(obj)
    print(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--same-row", action="store_true", help="with --build: every lane gathers the same p2 row (is layer 1 bound by the gather?)")
    ap.add_argument("--src", default=None, help="with --build: another version of fused_split.hip (its directory is searched for headers first)")
    ap.add_argument("--totals-only", action="store_true", help="with --build: only the whole-kernel clock / wall-clock counters (no phase ticks)")
    ap.add_argument("--out", default=SO, help="with --build: the library to write")
    ap.add_argument("-D", dest="defs", action="append", default=[], help="with --build: extra macro definitions (SP_F=24 ...)")
    ap.add_argument("--bwd", action="store_true", help="the backward kernel (cost_volume_bwd_split_kernel) instead of the forward one")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--so", default=SO, help="the instrumented library to run")
    a = ap.parse_args()
    if a.build:
        return build(a.same_row, a.src, a.totals_only, os.path.abspath(a.out), [os.path.dirname(os.path.abspath(a.src))] if a.src else [], a.defs, a.bwd)
    import ratrack_amd._lib as L
    L.SO_PATH = os.path.abspath(a.so)
    import torch
    if a.bwd:
        from ratrack_amd import train_ops as T
        ms, _ = T.time_cost_volume_bwd(a.batch, 256, "cuda", 20)
        print("cost volume backward B=%d alone (instrumented build): %.1f us per launch" % (a.batch, ms * 1e3))
    else:
        from ratrack_amd import synth
        from ratrack_amd.track4d import Args, Track4D
        net = Track4D(Args()).to("cuda").eval()
        synth.fill_state_dict(net.state_dict())
        net.invalidate_fused()
        d = synth.make_frame_pairs(a.batch, 256, 1)
        t = [torch.from_numpy(d[k]).to("cuda") for k in ("pc1", "pc2", "feature1", "feature2")]
        with torch.no_grad():
            net.backbone(*t, None)
            eng = net._fused_engine()
            eng.time_dominant_kernel(5)
            ev = eng.time_dominant_kernel(30)
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        print("cost volume forward B=%d alone (instrumented build): median %.1f us" % (a.batch, ms[len(ms) // 2] * 1e3))
    buf = (ctypes.c_ulonglong * (1024 * 16))()
    rc = L.load().rtk_dbg_cv_ticks(buf)
    assert rc == 0, rc
    import numpy as np
    tk = np.array(buf[:], dtype=np.float64).reshape(1024, 16)
    tk = tk[tk[:, 14] > 0]
    groups = 256 // 8
    gx = max(1, min(256 // a.batch, groups))
    tiles = groups / gx
    names = ["loop top", "layer 1 (rows through LDS, Wd.d, leaky)", "layer 2 (64 group steps of 6 MFMAs)", "a2 = leaky(acc c + b), position scale",
             "layer 3 (64 group steps)", "stream wrap sync", "epilogue: next index request, block 0 operands, WeightNet hidden layers",
             "epilogue: output block 0", "epilogue: output block 1", "epilogue: next tile's coordinates requested", "epilogue: output blocks 2..7",
             "epilogue: next direction (waits for the coordinates)", "position scale of a1 (max scan + swap)", "a3 = leaky(acc c + b)"]
    tot, wall = tk[:, 14].mean(), tk[:, 15].mean()
    print("%d waves, %.1f tiles each; kernel body %.0f clock ticks = %.0f wall ticks (100 MHz: %.1f us) -> clock runs at %.1f MHz"
          % (len(tk), tiles, tot, wall, wall / 100.0, tot / wall * 100.0))
    us = lambda x: x / tot * wall / 100.0
    if a.bwd:
        names = BWD_NAMES
    for k, nm in enumerate(names):
        print("  %-75s %9.0f ticks/tile  %6.2f us/tile  %5.1f %%   (min %.0f max %.0f over waves)"
              % (nm, tk[:, k].mean() / tiles, us(tk[:, k].mean() / tiles), 100 * tk[:, k].mean() / tot, tk[:, k].min() / tiles, tk[:, k].max() / tiles))
    print("  %-75s %9.0f ticks" % ("outside the tile loop (start_parts, first index, finish)", tot - tk[:, :len(names)].sum(1).mean()))
    print("  ideal MFMA time of a layer: 64 x 6 x 32 cycles = 12288 cycles")


if __name__ == "__main__":
    main()
