"""Same-box A/B of compile-time knobs in the pipelined forward (B = 64, N = 256, four batches in flight): variant libraries
(ratrack_amd/lib/variants), one process per run, base and variants alternating, `--rounds` times.
    python tools/experiments/ab_knobs.py --build (CPU)        python tools/experiments/ab_knobs.py [--rounds 2] (GPU)"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
VARIANTS = {      # tag -> (source file, extra flags); edit for the sweep at hand (HISTORY.md lists the ones that were run)
    "base": (None, []),
    "pw512": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=512"]), "pw128": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=128"]),
    "sa1280": ("fused_split.hip", ["-DSA_WGS_TARGET=1280"]), "san2048": ("fused_group.hip", ["-DSA_MAX_WGS=2048"]),
    "cv232": ("fused_split.hip", ["-DCV_FWD_VGPRS=232"]),
    "pw128": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=128"]), "pw192": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=192"]),
    "pw256": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=256"]), "pw320": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=320"]),
    "pw384": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=384"]), "pw448": ("fused_pointwise.hip", ["-DPW_WGS_TARGET=448"]),
}
if "--build" in sys.argv:
    from ratrack_amd import build as B
    B.build(verbose=False)
    os.makedirs(VAR, exist_ok=True)
    for tag, (fname, flags) in VARIANTS.items():
        objs = glob.glob(os.path.join(B.LIBDIR, "obj", "*.o"))
        extra = []
        if fname:
            src = os.path.join(B.CSRC, fname)
            obj = os.path.join(VAR, "%s_%s.o" % (fname[:-4], tag))
            subprocess.check_call([B._hipcc()] + B.flags_for(src) + flags + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", src, "-o", obj])
            objs = [o for o in objs if not o.endswith("/" + fname[:-4] + ".o")]
            extra = [obj]
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", os.path.join(VAR, "librtk_ab_%s.so" % tag)] + objs + extra)
        for o in extra:
            os.remove(o)
        print("built", tag, flush=True)
elif "--one" in sys.argv:
    from ratrack_amd import _lib
    _lib.SO_PATH = os.path.join(VAR, "librtk_ab_%s.so" % sys.argv[2])
    import torch
    from ratrack_amd import fused, synth
    from ratrack_amd.track4d import Args, Track4D
    dev = torch.device("cuda")
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    BB, NN = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 256)
    batches = []
    for i in range(8):
        d = synth.make_frame_pairs(BB, NN, 1000 + 100 * i)
        batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, BB, 128, device=dev)])
    with torch.no_grad():
        net.backbone(*batches[0])
        pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=4)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3000):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 3000 * 1e3), flush=True)
else:
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
    shape = sys.argv[sys.argv.index("--shape") + 1].split("x") if "--shape" in sys.argv else ["64", "256"]
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else list(VARIANTS)
    print("B=%s N=%s" % tuple(shape), flush=True)
    res = {t: [] for t in VARIANTS if t in only}
    for r in range(rounds):
        for tag in res:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", tag] + shape, capture_output=True, text=True)
            ms = [l for l in out.stdout.split("\n") if l.startswith("ONE ")]
            if ms:
                res[tag].append(float(ms[-1].split()[1]))
            else:
                print(tag, "FAILED", out.stderr[-300:], flush=True)
    b = sum(res["base"]) / max(len(res["base"]), 1)
    for tag, v in res.items():
        if v:
            m = sum(v) / len(v)
            print("%-8s %s  mean %.4f ms = %.1f k pairs/s  (%+.2f %% vs base)" % (tag, " ".join("%.4f" % x for x in v), m, int(shape[0]) / m, 100 * (b / m - 1)), flush=True)
