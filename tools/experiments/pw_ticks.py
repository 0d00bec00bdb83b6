#!/usr/bin/env python3
"""Where does a per-point chain launch (rtk_pointwise_mlp) spend its time?  The method of cv_ticks.py on fused_pointwise.hip: a
patched copy with clock reads at the phase boundaries, linked into ratrack_amd/lib/variants/librtk_pwticks.so.

    python tools/experiments/pw_ticks.py --build      (CPU)
    python tools/experiments/pw_ticks.py              (GPU: the forward's shapes one after the other)
"""
import argparse, ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
SO = os.path.join(VAR, "librtk_pwticks.so")
HEAD = r'''
__device__ unsigned long long g_pw_ticks[2048 * 8];
#define PW_TICK(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); tk[k] += t_ - tprev; tprev = t_; \
                     __builtin_amdgcn_sched_barrier(0); }
'''
TAIL = r'''
extern "C" __attribute__((visibility("default"))) int rtk_dbg_pw_ticks(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_ticks), sizeof(g_pw_ticks));
}
'''
PATCHES = [
    ("    WStream<PW_NW, PW_F, NF> ws;\n", "    unsigned long long tk[8] = {}, tprev = __builtin_readcyclecounter(), t00 = tprev;\n    WStream<PW_NW, PW_F, NF> ws;\n"),
    ("    for (int G = bx; G < live_groups; G += nbx) {\n        asm volatile(\"\" ::: \"memory\");   // keep loop-invariant",
     "    PW_TICK(0)\n    for (int G = bx; G < live_groups; G += nbx) {\n        asm volatile(\"\" ::: \"memory\");   // keep loop-invariant"),
    ("        if constexpr (NF > PW_F) ws.next();      // chunk 0", "        PW_TICK(1)\n        if constexpr (NF > PW_F) ws.next();      // chunk 0"),
    ("        if constexpr (SPLIT) mlp_layer_ws_split<U, V1, 0>(ws, h, a1);\n        else mlp_layer_ws<U, V1, 0>(ws, h, a1);\n",
     "        PW_TICK(2)\n        if constexpr (SPLIT) mlp_layer_ws_split<U, V1, 0>(ws, h, a1);\n        else mlp_layer_ws<U, V1, 0>(ws, h, a1);\n        PW_TICK(3)\n"),
    ("    }\n    ws.finish();\n}\n\ntemplate <int U, int V1, int V2, int V3, int V4>\nstatic int launch_pw",
     "        PW_TICK(4)\n    }\n    ws.finish();\n    tk[7] = __builtin_readcyclecounter() - t00;\n"
     "    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) g_pw_ticks[((blockIdx.x + gridDim.x * blockIdx.y) * PW_NW + (threadIdx.x >> 6)) * 8 + k] = tk[k];\n}\n\n"
     "template <int U, int V1, int V2, int V3, int V4>\nstatic int launch_pw"),
]
NAMES = ["before the tile loop (stream start)", "input loads issued (interpolation: index -> rows)", "stream entered: chunk 0 and the inputs have arrived",
         "first layer (weight stream)", "further layers, activation, store"]


def build():
    from ratrack_amd import build as B
    B.build(verbose=False)
    src = open(os.path.join(B.CSRC, "fused_pointwise.hip")).read()
    for anchor, repl in PATCHES:
        assert src.count(anchor) == 1, "anchor not unique / not found:\n" + anchor
        src = src.replace(anchor, repl)
    k = src.index("\n", src.index('#include "rtk_fused.h"')) + 1
    src = src[:k] + HEAD + src[k:] + TAIL
    os.makedirs(VAR, exist_ok=True)
    patched = os.path.join(VAR, "fused_pointwise_ticks.hip")
    open(patched, "w").write(src)
    obj = patched[:-4] + ".o"
    subprocess.check_call([B._hipcc()] + B.flags_for("fused_pointwise.hip") + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", patched, "-o", obj])
    objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if os.path.basename(o) != "fused_pointwise.o"]
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", SO] + objs + [obj])
    os.remove(obj)
    print(SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    a = ap.parse_args()
    if a.build:
        return build()
    import ratrack_amd._lib as L
    L.SO_PATH = SO
    import numpy as np
    import torch
    from ratrack_amd import fused as F
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    for S, rps, live, cin, couts in [(64, 256, 256, 128, [256]), (128, 512, 256, 128, [128]), (128, 512, 256, 96, [192]), (128, 512, 256, 64, [96]),
                                     (128, 512, 256, 128, [64]), (64, 256, 256, 128, [128, 64, 32, 3])]:
        rows = S * rps
        x = r(rows, cin)
        layers, ci = [], cin
        for co in couts:
            layers.append((r(co, ci).double() * 0.1, r(co).double(), F.ACT_RELU))
            ci = co
        chain = F.Chain(layers, dev)
        out = torch.empty(rows, ((couts[-1] + 15) // 16) * 16, device=dev)
        nu = torch.full((S,), live, dtype=torch.int32, device=dev)
        call = lambda: F.pointwise(rows, rps, [(x, cin, False)], chain, out, row_nuniq=nu)
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (2048 * 8))()
        assert L.load().rtk_dbg_pw_ticks(buf) == 0
        tk = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
        tk = tk[tk[:, 7] > 0]
        print("%d -> %s, %d x %d live rows: %.1f us per launch (back to back); %d waves, kernel body %.0f clocks (max %.0f)" %
              (cin, couts, S, live, e0.elapsed_time(e1) / 20 * 1e3, len(tk), tk[:, 7].mean(), tk[:, 7].max()))
        for k, nm in enumerate(NAMES):
            print("      %-62s %8.0f clocks  %5.1f %%" % (nm, tk[:, k].mean(), 100 * tk[:, k].mean() / tk[:, 7].mean()))
        macs, ci = 0, cin
        for co in couts:
            macs += ci * co; ci = co
        print("      MFMA time of the chain per 16-row tile: %d clocks (6 products x 16 clocks per 16x16x32)" % (6 * 16 * macs // (32 * 16)))


if __name__ == "__main__":
    main()
