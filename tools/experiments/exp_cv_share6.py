"""Round 6 re-check of the cost volume's share of the CUs in the pipelined forward (the other families got cheaper: does the best share move?).
One process per setting (hardware-queue lottery, see marginal_forward.py).  python tools/experiments/exp_cv_share6.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from ratrack_amd import fused, synth
    from ratrack_amd.track4d import Args, Track4D
    wgs, depth = int(sys.argv[2]), int(sys.argv[3])
    if wgs >= 0:
        fused.cv_shared_workgroups = lambda samples, n1, dev: wgs
    dev = torch.device("cuda")
    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    batches = []
    for i in range(8):
        d = synth.make_frame_pairs(64, 256, 1000 + 100 * i)
        batches.append([torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")] + [torch.zeros(5, 64, 128, device=dev)])
    with torch.no_grad():
        net.backbone(*batches[0])
        pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=depth)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 2000 * 1e3), flush=True)
    sys.exit(0)
for depth in (4, 3):
    for wgs in (-1, 128, 160, 176, 192, 208, 224, 240, 0):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(wgs), str(depth)], capture_output=True, text=True).stdout
        ms = float([l for l in out.split("\n") if l.startswith("ONE ")][-1].split()[1])
        print("depth %d, cost volume on %s workgroups: %.4f ms/batch = %.1f k pairs/s" % (depth, {-1: "the default (192)", 0: "all 256"}.get(wgs, wgs), ms, 64 / ms), flush=True)
