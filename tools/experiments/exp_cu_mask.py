"""Experiment: batch-level pipelining with CU-masked streams (hipExtStreamCreateWithCUMask): every batch in flight owns a
fixed share of the CUs, so latency-bound kernels do not occupy the whole chip and MFMA-bound ones do not collide."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ratrack_amd import synth, fused
from ratrack_amd.track4d import Track4D, Args

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def mask(part, nparts, mode):
    bits = 0
    for cu in range(256):
        if mode == "contig":
            own = cu * nparts // 256
        else:                       # interleaved: CU i -> partition i % nparts
            own = cu % nparts
        if own == part:
            bits |= 1 << cu
    return bits


dev = "cuda"; B = 64
net = Track4D(Args()).to(dev).eval(); synth.fill_state_dict(net.state_dict())
d = synth.make_frame_pairs(B, 256, 0); t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
h = torch.zeros(5, B, 128, device=dev)
inp = (t["pc1"], t["pc2"], t["feature1"], t["feature2"], h)
configs = [("none", 4, 0), ("contig", 4, 4), ("inter", 4, 4), ("contig", 4, 2), ("contig", 8, 4), ("contig", 6, 3), ("contig", 8, 8)]
for mode, depth, nparts in configs:
    with torch.no_grad():
        eng = fused.FusedBackbone(net)
        pipe = fused.GraphPipeline(eng, inp, depth=depth)
        if nparts:
            pipe.streams = [masked_stream(mask(k % nparts, nparts, mode)) for k in range(depth)]
        else:
            pipe.streams = pipe.streams[:depth]
        for _ in range(12): pipe.submit(*inp)
        pipe.drain(); torch.cuda.synchronize()
        n = 240
        t0 = time.perf_counter()
        for _ in range(n): pipe.submit(*inp)
        pipe.drain(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    print("mask %-7s depth %d partitions %d: %.4f ms/step  %.0f pairs/s" % (mode, depth, nparts, ms, B / ms * 1e3), flush=True)
