"""Upper bound of folding the level transitions (sa out -> linear || next projections: t1, t2, l3 of both PNHeads, six launches) into the
SA kernels' epilogues: the pipelined forward with those six launches skipped (results wrong), one process per setting.
python tools/experiments/exp_skip_transitions.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from ratrack_amd import fused, synth
    from ratrack_amd.track4d import Args, Track4D
    mode = sys.argv[2]
    dev = torch.device("cuda")
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    batches = []
    for i in range(8):
        d = synth.make_frame_pairs(64, 256, 1000 + 100 * i)
        batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)])
    with torch.no_grad():
        net.backbone(*batches[0])
        eng = net._fused
        skip = set()
        if mode in ("transitions", "both"):
            for W in (eng.enc, eng.dec):
                skip |= {id(W.trans[0]), id(W.trans[1]), id(W.lin3)}
        if mode in ("heads", "both"):
            skip |= {id(eng.cls_head), id(eng.dec_q1)}
        if mode == "p12":
            skip |= {id(eng.p1_loc), id(eng.p2_loc)}
        if mode == "fp":
            for W in (eng.enc, eng.dec):
                skip |= {id(W.fp["fp3"]), id(W.fp["fp2"])}
        orig = fused.pointwise

        def pw(rows, rps, srcs, chain, out, *a, **k):
            if id(chain) in skip:
                return out
            return orig(rows, rps, srcs, chain, out, *a, **k)
        fused.pointwise = pw
        pipe = fused.GraphPipeline(eng, tuple(batches[0]), depth=4)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 2000 * 1e3), flush=True)
    sys.exit(0)
for mode in ("none", "p12", "fp", "transitions", "heads", "none", "p12", "fp"):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", mode], capture_output=True, text=True).stdout
    ms = float([l for l in out.split("\n") if l.startswith("ONE ")][-1].split()[1])
    print("skipped: %-12s %.4f ms/batch = %.1f k pairs/s" % (mode, ms, 64 / ms), flush=True)
