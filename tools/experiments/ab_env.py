"""Same-box A/B of runtime environment settings for the pipelined forward (B = 64, N = 256, depth 4 unless given): one process per run.
python tools/experiments/ab_env.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--one" in sys.argv:
    depth = int(sys.argv[2])
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
        pipe = fused.GraphPipeline(net._fused, tuple(batches[0]), depth=depth)
        for i in range(400):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3000):
            pipe.submit(*batches[i % 8])
        pipe.drain(); torch.cuda.synchronize()
        print("ONE %.4f" % ((time.perf_counter() - t0) / 3000 * 1e3), flush=True)
    sys.exit(0)
CASES = [("default", {}, 4), ("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}, 4), ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}, 4),
         ("GPU_MAX_HW_QUEUES=5 depth 5", {"GPU_MAX_HW_QUEUES": "5"}, 5), ("GPU_MAX_HW_QUEUES=6 depth 6", {"GPU_MAX_HW_QUEUES": "6"}, 6),
         ("GPU_MAX_HW_QUEUES=8 depth 4", {"GPU_MAX_HW_QUEUES": "8"}, 4), ("GPU_MAX_HW_QUEUES=8 depth 8", {"GPU_MAX_HW_QUEUES": "8"}, 8),
         ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}, 4), ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}, 4)]
res = {c[0]: [] for c in CASES}
for rep in range(2):
    for name, env, depth in CASES:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(depth)], capture_output=True, text=True, env=dict(os.environ, **env))
        ms = [l for l in out.stdout.split("\n") if l.startswith("ONE ")]
        if ms:
            res[name].append(float(ms[-1].split()[1]))
        else:
            print(name, "FAILED", out.stderr[-200:], flush=True)
b = sum(res["default"]) / max(len(res["default"]), 1)
for name, v in res.items():
    if v:
        m = sum(v) / len(v)
        print("%-30s %s  mean %.4f ms = %.1f k pairs/s (%+.2f %%)" % (name, " ".join("%.4f" % x for x in v), m, 64 / m, 100 * (b / m - 1)), flush=True)
