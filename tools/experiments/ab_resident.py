"""Same-box A/B of the resident-image per-point layers (r06_pointwise_resident.patch applied to the working tree): variant libraries with
different workgroup shapes, python flag fused.PW_RESIDENT on / off.  --build (CPU), then run on the GPU."""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
LIBS = {"nw4_256": ["-DPWR_NW=4", "-DPWR_WGS_TARGET=256"], "nw4_512": ["-DPWR_NW=4", "-DPWR_WGS_TARGET=512"],
        "nw8_256": ["-DPWR_NW=8", "-DPWR_WGS_TARGET=256"], "nw8_128": ["-DPWR_NW=8", "-DPWR_WGS_TARGET=128"]}
if "--build" in sys.argv:
    from ratrack_amd import build as B
    B.build(verbose=False)
    os.makedirs(VAR, exist_ok=True)
    src = os.path.join(B.CSRC, "fused_pointwise.hip")
    for tag, flags in LIBS.items():
        obj = os.path.join(VAR, "pw_%s.o" % tag)
        subprocess.check_call([B._hipcc()] + B.flags_for(src) + flags + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", src, "-o", obj])
        objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if not o.endswith("/fused_pointwise.o")]
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", os.path.join(VAR, "librtk_res_%s.so" % tag)] + objs + [obj])
        os.remove(obj)
        print("built", tag, flush=True)
elif "--one" in sys.argv:
    from ratrack_amd import _lib
    _lib.SO_PATH = os.path.join(VAR, "librtk_res_%s.so" % sys.argv[2])
    import torch
    from ratrack_amd import fused, synth
    from ratrack_amd.track4d import Args, Track4D
    fused.PW_RESIDENT = sys.argv[3] == "1"
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
        for i in range(3000):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 3000 * 1e3), flush=True)
else:
    runs = [("nw4_256", "0")] + [(t, "1") for t in LIBS]
    res = {r: [] for r in runs}
    for rep in range(2):
        for r in runs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", r[0], r[1]], capture_output=True, text=True)
            ms = [l for l in out.stdout.split("\n") if l.startswith("ONE ")]
            if ms:
                res[r].append(float(ms[-1].split()[1]))
            else:
                print(r, "FAILED", out.stderr[-400:], flush=True)
    b = sum(res[runs[0]]) / max(len(res[runs[0]]), 1)
    for r, v in res.items():
        if v:
            m = sum(v) / len(v)
            print("%-8s resident=%s  %s  mean %.4f ms = %.1f k pairs/s (%+.2f %%)" % (r[0], r[1], " ".join("%.4f" % x for x in v), m, 64 / m, 100 * (b / m - 1)), flush=True)
