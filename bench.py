#!/usr/bin/env python3
"""Headline benchmark: radar frame-pairs/s of the RaTrack backbone forward (eval) at B=64, N=256 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one Track4D.backbone() pass (PointNet++ MSG encoder over both frames, kNN cost volume,
decoder PNHead, GRU, scene-flow + motion-segmentation heads) over one synthetic batch of B frame-pairs
already resident in HBM.  One process per GPU; frame-pairs are independent, so ranks run disjoint
batches with no data-path collective (weak scaling); the timed region is bracketed by a barrier +
synchronize on both sides and the maximum over ranks is reported.

`--mode train` times the training step instead (BASELINE configs 3/4: forward + multi-task loss + backward + gradient
all-reduce + Adam); its roofline entry is the cost-volume backward kernel, its CPU baseline the oracle's train step.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (cost_volume_kernel) against the fp32 MFMA peak, duration measured
                live with HIP events on the launch stream during the timed region;
  cpu_baseline  the CPU oracle (oracle/track4d_ref.py: C restatement of the native ops + PyTorch-CPU
                dense layers) timed on this box's host cores on a bounded sample (N=1 run only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per frame-pair forward (fp32, weights excluded)
ALG_BYTES_PER_PAIR = {256: 14154240, 1024: 25293312}
ALG_FLOPS_PER_PAIR = {256: 4.003e9, 1024: 10.790e9}
EXEC_FLOPS_PER_PAIR = {256: 1.78e9}
FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def cost_volume_flops_per_pair(n, k=16):
    """Work of the stage the dominant kernel implements, counted in the reference's own arithmetic
    (model_utils.py:226-236; SURVEY.md Appendix B 'stage 1' minus the layer-1 feature part, which this
    design evaluates per point in separate launches): per (point, neighbour) pair
    layers 2+3 (2 x 256 x 256) + layer-1 direction term (3 x 256) + WeightNet (2136) + weighted sum (256) MACs."""
    macs = k * n * (2 * 256 * 256 + 3 * 256 + 2136 + 256)
    return 2.0 * macs


def cpu_baseline(batch, n, budget_s=20.0):
    from oracle import track4d_ref as R
    from ratrack_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args())
    synth.fill_state_dict(net.state_dict())
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    d = synth.make_frame_pairs(batch, n, case_id=99)
    t = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
    with torch.no_grad():
        R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)       # warm-up
        runs, t0 = 0, time.perf_counter()
        while True:
            R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
            runs += 1
            el = time.perf_counter() - t0
            if el > budget_s or runs >= 20:
                break
    return {"value": round(batch * runs / el, 3), "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d forward passes of the CPU oracle at B=%d, N=%d (%.1f s)" % (runs, batch, n, el)}


def time_cost_volume_bwd(batch, n, dev, iters=10):
    """Dominant kernel of the train step (cost_volume_bwd_kernel: forward recompute + both 256x256 input gradients of the
    cost volume, 4 x 2 x 256 x 256 MACs per (point, neighbour) pair) timed live with HIP events at the bench shape."""
    from ratrack_amd import _lib, train_ops
    from ratrack_amd.fused import pack_layer
    g = torch.Generator(dev).manual_seed(0)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    B, M = batch, batch * n * 16
    xyz1, xyz2 = r(B, n, 3).contiguous(), r(B, n, 3).contiguous()
    knn = torch.randint(0, n, (B, n, 16), device=dev, generator=g)
    p1, p2, dout = r(B * n, 256), r(B * n, 256), r(B * n, 256)
    w2, w3 = r(256, 256) * 0.06, r(256, 256) * 0.06
    W = train_ops._CvWeights(r(256, 3), w2, r(256), w3, r(256), r(8, 3), r(8), r(8, 8), r(8), r(256, 8), r(256), backward=True)
    wct = pack_layer(r(8, 256))
    big = torch.empty(6, M, 256, device=dev)
    d4, dt2 = torch.empty(M, 4, device=dev), torch.empty(M, 8, device=dev)
    dp1, dpd = torch.empty(B * n, 256, device=dev), torch.empty(B * n, 3, 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.call("rtk_cost_volume_bwd", B, n, n, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), p1.data_ptr(), p2.data_ptr(),
                  W.wd.data_ptr(), W.layers, W.wn, wct.data_ptr(), dout.data_ptr(), 256, 256, big[0].data_ptr(), big[1].data_ptr(),
                  big[2].data_ptr(), big[3].data_ptr(), big[4].data_ptr(), big[5].data_ptr(), d4.data_ptr(), dp1.data_ptr(),
                  dpd.data_ptr(), dt2.data_ptr(), st)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * (4 * 256 * 256 + 3 * 256 + 2136 + 256 + 8 * 256)
    return ms, flops


def cpu_train_baseline(batch, n, budget_s=20.0):
    """The CPU oracle's train step (train-mode forward + multi-task loss + backward, torch-CPU autograd over the C ops)."""
    from oracle import track4d_ref as R
    from ratrack_amd import loss as L
    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args())
    synth.fill_state_dict(net.state_dict())
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in net.state_dict().items()}
    d = synth.make_frame_pairs(batch, n, case_id=98)
    t = {k: torch.from_numpy(v) for k, v in d.items()}
    runs, t0 = 0, time.perf_counter()
    while True:
        flow, h, cls, *_ = R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None, training=True)
        total, _ = L.backbone_loss(t["pc1"] + flow, cls, t["gt_warp"], t["gt_cls"])      # a few reductions, device-agnostic
        total.backward()
        runs += 1
        el = time.perf_counter() - t0
        if el > budget_s or runs >= 10:
            break
    return {"value": round(batch * runs / el, 3), "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps (forward + loss + backward) of the CPU oracle at B=%d, N=%d (%.1f s)" % (runs, batch, n, el)}


def bench_train(a, net, d, dev, dist, world, rank):
    """Train step per rank on B frame-pairs: train-mode forward (HIP ops + PyTorch-ROCm dense layers, autograd),
    multi-task loss, backward, ONE flat gradient all-reduce over RCCL, Adam.  Weak scaling (B per GPU fixed)."""
    from ratrack_amd.ddp import broadcast_parameters
    from ratrack_amd.train import Trainer
    broadcast_parameters(net)
    # whole-step capture is validated on one GPU; with a collective inside (RCCL all-reduce under stream capture) it is opt-in
    # until it has been run on a multi-GPU node
    use_graph = (not a.no_graph) and (world == 1 or os.environ.get("RTK_TRAIN_GRAPH_DDP") == "1")
    tr = Trainer(net, graph=use_graph)
    t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    h = torch.zeros(5, a.batch, 128, device=dev)
    step = lambda: tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        pairs = a.batch * world * a.steps / el
        kms, kflops = time_cost_volume_bwd(a.batch, a.npoints, dev)
        ach = kflops / (kms * 1e-3) / 1e12
        traffic = None
        try:   # fabric-side bytes per launch from the committed PMC passes (same workload only)
            if a.batch == 64 and a.npoints == 256:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_cost_volume_bwd.json")))["traffic_bytes_per_launch"]
        except Exception:
            pass
        roof = {"kernel": "cost_volume_bwd_kernel", "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": traffic, "kernel_ms": round(kms, 4), "flops_per_launch": kflops,
                "share_of_step": round(kms / (el / a.steps * 1e3), 3)}
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            try:
                cpu = cpu_train_baseline(min(a.batch, 4), a.npoints)
            except Exception as e:          # the baseline is a reported extra, never a reason to lose the measurement
                cpu = {"error": repr(e)[:200]}
        print(json.dumps({
            "metric": "radar frame-pairs/sec (train step) at B=%d,N=%d per GPU" % (a.batch, a.npoints), "value": round(pairs, 1),
            "unit": "frame-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Track4D.backbone train step (fwd+loss+bwd+grad all-reduce+Adam), B=%d x N=%d per GPU, hipGraph=%s"
                                   % (a.batch, a.npoints, use_graph), "global_batch": a.batch * world,
                       "parallelism": "dp%d, one flat RCCL all-reduce of %d bytes per step" % (
                           world, tr.reducer.payload_bytes or 4 * sum(p.numel() for p in net.parameters() if p.grad is not None))},
            "roofline": roof, "cpu_baseline": cpu}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frame-pairs per GPU per step")
    ap.add_argument("--npoints", type=int, default=256, help="radar points per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--pipeline", type=int, default=2, help="captured graphs in flight (batch-level pipelining on streams)")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward = the headline metric (eval backbone, fused kernels); train = forward+loss+backward+"
                         "gradient all-reduce+Adam on the training path, captured in one hipGraph (BASELINE config 3/4), same fields")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)      # RCCL
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world)

    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D

    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    net.invalidate_fused()
    d = synth.make_frame_pairs(a.batch, a.npoints, case_id=1000 + rank)          # every rank its own batch
    pc1, pc2 = torch.from_numpy(d["pc1"]).to(dev), torch.from_numpy(d["pc2"]).to(dev)
    f1, f2 = torch.from_numpy(d["feature1"]).to(dev), torch.from_numpy(d["feature2"]).to(dev)
    h = torch.zeros(5, a.batch, 128, device=dev)

    if a.mode == "train":
        return bench_train(a, net, d, dev, dist, world, rank)

    eng = None
    with torch.no_grad():
        out = net.backbone(pc1, pc2, f1, f2, h)
        eng = net._fused
        assert eng, "fused engine not active"
        step = lambda: net.backbone(pc1, pc2, f1, f2, h)
        pipe = None
        if not a.no_graph:
            from ratrack_amd.fused import GraphPipeline
            pipe = GraphPipeline(eng, (pc1, pc2, f1, f2, h), depth=max(1, a.pipeline))
            step = lambda: pipe.submit(pc1, pc2, f1, f2, h)        # inputs are copied into the slot's static buffers

        def barrier():
            if pipe is not None:
                pipe.drain()                                       # every submitted batch finishes inside the timed region
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(a.warmup):
            step()
        eng.kernel_events = []                      # (start, stop) HIP events around the dominant kernel
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        events, eng.kernel_events = eng.kernel_events, None
        if not events:                              # graph replay: time the dominant kernel on the same stream right after
            events = eng.time_dominant_kernel(20)

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_ms = sum(s.elapsed_time(e) for s, e in events) / max(len(events), 1)

    if rank == 0:
        pairs_per_s = a.batch * world * a.steps / elapsed
        cv_flops = cost_volume_flops_per_pair(a.npoints) * a.batch
        achieved = cv_flops / (kern_ms * 1e-3) / 1e12
        traffic = None
        try:   # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (same workload only)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_cost_volume.json")))
            if a.batch == 64 and a.npoints == 256:
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        res = {
            "metric": "radar frame-pairs/sec (backbone forward, eval) at B=%d,N=%d per GPU" % (a.batch, a.npoints),
            "value": round(pairs_per_s, 1),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Track4D.backbone forward, B=%d frame-pairs x N=%d points per GPU, S=512 centroids, "
                                   "eval-mode BN, random-init weights, hipGraph=%s, batches in flight=%d"
                                   % (a.batch, a.npoints, not a.no_graph, 1 if a.no_graph else max(1, a.pipeline)),
                       "global_batch": a.batch * world, "parallelism": "replicas x%d (no collective on the forward path)" % world},
            "roofline": {"kernel": "cost_volume_kernel", "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms, 4), "flops_per_launch": cv_flops},
            # whole path per GPU against both rooflines (SURVEY.md H1 asks for both).  "algorithmic" = the reference
            # formulation's 14.15 MB / 4.003 GFLOP per pair; "executed" = the multiply-adds this design actually issues
            # (per-point layer-1 projections, duplicate centroids skipped): 1.78 GFLOP per pair at N=256.
            "whole_path": {"hbm_frac_algorithmic": round(pairs_per_s / world * ALG_BYTES_PER_PAIR.get(a.npoints, 0) / (HBM_PEAK_GBS * 1e9), 5),
                           "fp32_frac_algorithmic": round(pairs_per_s / world * ALG_FLOPS_PER_PAIR.get(a.npoints, 0) / (FP32_PEAK_TFLOPS * 1e12), 5),
                           "fp32_frac_executed": round(pairs_per_s / world * EXEC_FLOPS_PER_PAIR.get(a.npoints, 0) / (FP32_PEAK_TFLOPS * 1e12), 5)},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(8, a.npoints)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
