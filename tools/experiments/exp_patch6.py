"""Is rtk_patch_cost bound by its gathers?  Patched copies of csrc/fused_group.hip (results of the ablations are wrong), each linked into
ratrack_amd/lib/variants/librtk_patch_<tag>.so and timed alone at the bench shape (B = 64, N = 256, real kNN tables).
    python tools/experiments/exp_patch6.py --build   (CPU)      python tools/experiments/exp_patch6.py   (GPU)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "ratrack_amd", "lib", "variants")
VARIANTS = {
    "base": [],
    # every neighbour is the point itself: 16 lanes read the same row (perfect locality, same instruction stream)
    "self": [("        const long nb = (long)b * P.n + (long)P.knn[i * 16 + j];\n        const float bop = g < 3 ? __fsub_rn(P.xyz[nb * 3 + g], P.xyz[i * 3 + g]) : 1.0f;\n        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);\n        const float *fr = P.feat + nb * P.feat_pitch + 4 * g;\n        // four 16-channel",
              "        const long nb = i + 0 * (long)P.knn[i * 16 + j];\n        const float bop = g < 3 ? __fsub_rn(P.xyz[nb * 3 + g], P.xyz[i * 3 + g]) : 1.0f;\n        const f4 t2 = weightnet_hidden(P.wn, lane, g, bop);\n        const float *fr = P.feat + nb * P.feat_pitch + 4 * g;\n        // four 16-channel")],
}


def build():
    from ratrack_amd import build as B
    B.build(verbose=False)
    src0 = open(os.path.join(B.CSRC, "fused_group.hip")).read()
    os.makedirs(VAR, exist_ok=True)
    for tag, patches in VARIANTS.items():
        src = src0
        for a, b in patches:
            assert src.count(a) == 1, (tag, src.count(a))
            src = src.replace(a, b)
        path = os.path.join(VAR, "fused_group_%s.hip" % tag)
        open(path, "w").write(src)
        obj = path[:-4] + ".o"
        subprocess.check_call([B._hipcc()] + B.flags_for(os.path.join(B.CSRC, "fused_group.hip")) + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", path, "-o", obj])
        objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if not o.endswith("/fused_group.o")]
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", os.path.join(VAR, "librtk_patch_%s.so" % tag)] + objs + [obj])
        os.remove(obj); os.remove(path)
        print("built", tag)


if "--build" in sys.argv:
    build()
elif "--one" in sys.argv:
    from ratrack_amd import _lib
    _lib.SO_PATH = os.path.join(VAR, "librtk_patch_%s.so" % sys.argv[2])
    import torch
    from ratrack_amd import fused as F, benchutil as BU, synth, pointnet2_utils as PU
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args()).to("cuda").eval(); synth.fill_state_dict(net.state_dict())
    eng = F.FusedBackbone(net)
    B, N = 64, 256
    d = synth.make_frame_pairs(B, N, 1000)
    xyz = torch.from_numpy(d["pc1"]).permute(0, 2, 1).contiguous().cuda()
    knn = PU.knn_point(16, xyz, xyz)
    feat = torch.randn(B * N, 256, device="cuda"); out = torch.empty(B * N, 256, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("rtk_patch_cost", B, N, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, eng.wn2.arr, out.data_ptr(), 256, 0, st())
    print("ONE %-6s patch_cost %.1f us" % (sys.argv[2], BU.time_graph(f, 20) * 1e3), flush=True)
else:
    for tag in VARIANTS:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", tag], capture_output=True, text=True)
        print([l for l in out.stdout.split("\n") if l.startswith("ONE")] or out.stderr[-400:], flush=True)
