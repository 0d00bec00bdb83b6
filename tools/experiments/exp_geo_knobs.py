"""Compile-time knobs of the round-6 geometry kernels in the pipelined forward: workgroups of the tables launch, lanes per query of the
selection kernels, centroids per wave of the ball queries.  Variant libraries (ratrack_amd/lib/variants), one process per setting.
    python tools/experiments/exp_geo_knobs.py --build (CPU)        python tools/experiments/exp_geo_knobs.py (GPU)"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
VARIANTS = {"base": [], "tables1024": ["-DGEO_TABLES_WGS=1024"], "tables4096": ["-DGEO_TABLES_WGS=4096"], "tables8192": ["-DGEO_TABLES_WGS=8192"],
            "lpq8": ["-DKV_LPQ=8"], "lpq16": ["-DKV_LPQ=16"], "bq1": ["-DBQ_CENTROIDS_PER_WAVE=1"], "bq4": ["-DBQ_CENTROIDS_PER_WAVE=4"]}
if "--build" in sys.argv:
    from ratrack_amd import build as B
    B.build(verbose=False)
    os.makedirs(VAR, exist_ok=True)
    src = os.path.join(B.CSRC, "ops_pointnet2.hip")
    for tag, flags in VARIANTS.items():
        obj = os.path.join(VAR, "ops_%s.o" % tag)
        subprocess.check_call([B._hipcc()] + B.flags_for(src) + flags + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", src, "-o", obj])
        objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if not o.endswith("/ops_pointnet2.o")]
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", os.path.join(VAR, "librtk_geo_%s.so" % tag)] + objs + [obj])
        os.remove(obj)
        print("built", tag, flush=True)
elif "--one" in sys.argv:
    from ratrack_amd import _lib
    _lib.SO_PATH = os.path.join(VAR, "librtk_geo_%s.so" % sys.argv[2])
    import torch
    from ratrack_amd import fused, synth
    from ratrack_amd.track4d import Args, Track4D
    dev = torch.device("cuda")
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    batches = []
    for i in range(8):
        d = synth.make_frame_pairs(64, 256, 1000 + 100 * i)
        batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)])
    with torch.no_grad():
        net.backbone(*batches[0])
        pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=4)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2500):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 2500 * 1e3), flush=True)
else:
    for rep in range(2):
        for tag in VARIANTS:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", tag], capture_output=True, text=True)
            ms = [l for l in out.stdout.split("\n") if l.startswith("ONE ")]
            print("%-12s %s" % (tag, ("%.4f ms/batch = %.1f k pairs/s" % (float(ms[-1].split()[1]), 64 / float(ms[-1].split()[1]))) if ms else out.stderr[-300:]), flush=True)
