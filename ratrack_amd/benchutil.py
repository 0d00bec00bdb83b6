"""Measurement helpers of bench.py (no product logic): live HIP-event timings of the irregular (geometry / gather / scatter)
kernels at the bench shape with their algorithmic byte counts -- north_star asks for these ops' GB/s against the 8 TB/s
HBM roofline ("HBM bandwidth for n1-n6, the kNN and every gather/scatter")."""
import torch

from . import _lib, fused
from . import pointnet2_hip as _native

HBM_PEAK_GBS = 8000.0


def _time(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters          # ms per launch, on the launch stream


def time_graph(fn, reps=20, replays=5):
    """ms per launch from a replayed hipGraph of `reps` launches of fn (an eager ctypes launch costs ~10 us: too coarse for the
    microsecond kernels of the training path)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (replays * reps)


def irregular_ops(batch, n, dev, npoint=512, iters=20, pmc=None):
    """One entry per irregular kernel of a forward / train step at B=batch frame-pairs of n points (2B clouds through the
    shared encoder): launches per step, ms per launch (HIP events, back-to-back launches on the current stream), the
    ALGORITHMIC bytes per launch (compulsory reads + writes of the rows the kernel computes; duplicate centroids beyond the
    exhausted-cloud counter are not computed), GB/s and the fraction of the 8 TB/s HBM peak.  `pmc`: optional
    {kernel: fabric bytes per launch} from the committed PMC passes (profiles/r02_irregular_hbm.json) -> "traffic"."""
    from . import synth
    d = synth.make_frame_pairs(batch, n, case_id=1000)
    xyz = torch.cat([torch.from_numpy(d["pc1"]), torch.from_numpy(d["pc2"])]).permute(0, 2, 1).contiguous().to(dev)
    S_ = 2 * batch
    st = fused._stream
    geo = fused.Geometry(xyz, npoint, side=None, knn_frames=batch)
    torch.cuda.synchronize()
    U = [float(c.double().mean().item()) for c in geo.nuniq]            # live centroid rows per sample and level
    out = []

    def add(kernel, per_step, ms, nbytes, what):
        gbs = nbytes / (ms * 1e-3) / 1e9
        e = {"kernel": kernel, "launches_per_step": per_step, "ms": round(ms, 5), "bytes": int(nbytes), "GB/s": round(gbs, 1),
             "frac": round(gbs / HBM_PEAK_GBS, 5), "what": what}
        if pmc and kernel in pmc:
            e["traffic"] = pmc[kernel]
        out.append(e)

    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    # ---- the product path's geometry: two launches (round 6), timed where the backbone issues them ---------------------------------
    import statistics
    api = [torch.from_numpy(d[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")]
    q1w = torch.randn(32, 2, device=dev)
    per = {"rtk_geometry_front": [], "rtk_geometry_tables": []}
    for _ in range(9):
        _lib.TIMING = rec = []
        g2 = fused.Geometry(f32(S_, n, 3), npoint, side=None, knn_frames=batch, prepare=(*api, f32(S_ * n, 4)), q1=(q1w, f32(S_ * n, 32)))
        _lib.TIMING = None
        torch.cuda.synchronize()
        for nm, e0, e1 in rec:
            if nm in per:
                per[nm].append(e0.elapsed_time(e1))
        del g2
    if per["rtk_geometry_front"]:
        ns_all = [ns for row in fused._PNHeadWeights.NSAMPLES for ns in row]
        front = S_ * (n * 20 + n * (12 + 16 + 128) + sum(U) * 16 + 24) + 2 * batch * n * 16 * 8
        tables = S_ * (n * 12 + 2 * sum(U) * 12 + sum(U[l] * 4 * (ns_all[2 * l] + ns_all[2 * l + 1]) for l in range(3)) + (U[1] + U[0] + n) * 24)
        add("geometry_front_kernel", 1, statistics.median(per["rtk_geometry_front"]), front,
            "layout conversion + sa1 projection of the raw features + FPS levels 1-3 (one wave per cloud) + both kNN tables, %d clouds" % S_)
        add("geometry_tables_kernel", 1, statistics.median(per["rtk_geometry_tables"]), tables,
            "six ball queries + three three-NN tables over the unique centroids, %d clouds" % S_)
    # ---- furthest point sampling + centroid gather, level 1 (the 511-round dependent chain) ----------------------------
    idx, nx, cnt, tie = i32(S_, npoint), f32(S_, npoint, 3), i32(S_), i32(S_)
    tie23, first, snap = i32(2, S_), i32(S_), f32(S_, n)
    ms = _time(lambda: _lib.call("rtk_fps_centroids", S_, n, npoint, xyz.data_ptr(), idx.data_ptr(), nx.data_ptr(), cnt.data_ptr(),
                                 tie.data_ptr(), None, snap.data_ptr(), first.data_ptr(), st()), iters)
    add("fps_wave_kernel", 1, ms, S_ * (n * 12 + npoint * 16 + 8), "FPS %d -> %d + centroid gather, %d clouds" % (n, npoint, S_))
    idx23, nx23, cnt23 = i32(2, S_, npoint), f32(2, S_, npoint, 3), i32(2, S_)
    ms = _time(lambda: _lib.call("rtk_fps_relevel", S_, npoint, 2, nx.data_ptr(), cnt.data_ptr(), tie.data_ptr(), idx23.data_ptr(),
                                 nx23.data_ptr(), cnt23.data_ptr(), tie23.data_ptr(), idx.data_ptr(), snap.data_ptr(), n, first.data_ptr(), st()), iters)
    add("fps_wave_kernel (levels 2, 3)", 2, ms, S_ * (npoint * 12 + 2 * npoint * 16 + 12), "levels 2, 3 (copy unless a level-1 tie)")
    # ---- ball queries: both scales of a level in one scan ----------------------------------------------------------------
    for lvl in range(3):
        (r1, r2), (n1, n2) = fused._PNHeadWeights.RADII[lvl], fused._PNHeadWeights.NSAMPLES[lvl]
        src, dst = geo.xyz[lvl], geo.xyz[lvl + 1]
        b1, b2 = geo.ball[lvl]
        ms = _time(lambda: _lib.call("rtk_ball_query_pair", S_, src.shape[1], npoint, float(r1), n1, float(r2), n2, dst.data_ptr(),
                                     src.data_ptr(), b1.data_ptr(), b2.data_ptr(), geo.nuniq[lvl].data_ptr(), st()), iters)
        nsrc = src.shape[1] if lvl == 0 else U[lvl - 1]
        add("ball_query_pair_kernel[l%d]" % (lvl + 1), 1, ms, S_ * (nsrc * 12 + U[lvl] * (12 + 4 * (n1 + n2))),
            "r=(%g,%g) ns=(%d,%d), %d source points" % (r1, r2, n1, n2, src.shape[1]))
    # ---- three nearest neighbours ----------------------------------------------------------------------------------------
    for name, (u, k) in {"fp3": (2, 3), "fp2": (1, 2), "fp1": (0, 1)}.items():
        d2, ix, m = geo.nn[name]
        nu = geo.xyz[u].shape[1]
        mask = geo.nuniq[u - 1].data_ptr() if u > 0 else None
        ms = _time(lambda: _lib.call("rtk_three_nn_masked", S_, nu, m, geo.xyz[u].data_ptr(), geo.xyz[k].data_ptr(), d2.data_ptr(),
                                     ix.data_ptr(), mask, geo.nuniq[k - 1].data_ptr(), st()), iters)
        rows = nu if u == 0 else U[u - 1]
        add("three_nn_kernel[%s]" % name, 1, ms, S_ * (rows * 12 + U[k - 1] * 12 + rows * 24), "%d unknown vs %d known" % (nu, m))
    # ---- kNN (k = 16) of the cost volume: torch.topk-based knn_point in the reference ---------------------------------------
    x1, x2 = xyz[:batch], xyz[batch:]
    knn = torch.empty(batch, n, 16, dtype=torch.int64, device=dev)
    ms = _time(lambda: _native.knn_point_wrapper(batch, n, n, 16, x1, x2, knn), iters)
    add("knn_point_kernel", 2, ms, batch * (2 * n * 12 + n * 16 * 8), "k=16, %d x %d per pair, int64 indices" % (n, n))
    # ---- training-side gather gradients ------------------------------------------------------------------------------------
    M = batch * n * 16
    src_rows = torch.randn(M, 256, device=dev)
    dst_rows = f32(batch * n, 256)
    from . import train_ops  # noqa: F401  (registers the signatures)
    ms = _time(lambda: _lib.call("rtk_scatter_add_rows", batch, n * 16, n, 256, knn.data_ptr(), src_rows.data_ptr(), dst_rows.data_ptr(),
                                 st()), iters)
    add("scatter_add_rows_kernel", 2, ms, M * 256 * 4 + M * 8 + batch * n * 256 * 4,
        "gather backward of the cost volume: %d rows x 256 ch -> %d rows, atomic-free" % (M, batch * n))
    # first-layer backward of the largest set-abstraction scale (level 3, 64 channels x 32 neighbours): gather form over the inverse
    # index (the reference scatters with one atomicAdd per element, group_points_gpu.cu:8-25)
    from .train_path import TrainGeometry
    tg = TrainGeometry(xyz, npoint)
    idx3, dxyz3 = tg.ball[2][1], tg.dxyz[2][1]
    rows, ns, C = idx3.shape[1], idx3.shape[2], 64
    off, inv = tg.inv[2][1]
    ms = _time(lambda: _lib.call("rtk_group_inverse_index", S_, rows, rows * ns, idx3.data_ptr(), off.data_ptr(), inv.data_ptr(), st()), iters)
    add("inverse_index_kernel", 6, ms, S_ * (rows * ns * 4 + rows * ns * 2 + (rows + 1) * 4),
        "positions sorted by gathered source row, largest table: %d centroids x %d neighbours" % (rows, ns))
    dz, dproj, dwx = torch.randn(S_, C, rows, ns, device=dev), f32(S_, C, rows), torch.zeros(C, 3, device=dev)
    dwx_ws = f32(S_ * C, 3)
    ms = _time(lambda: _lib.call("rtk_sa_first_layer_bwd", S_, C, rows, ns, rows, dz.data_ptr(), dxyz3.data_ptr(), off.data_ptr(),
                                 inv.data_ptr(), dproj.data_ptr(), dwx.data_ptr(), 3, dwx_ws.data_ptr(), st()), iters)
    add("sa_first_layer_bwd_kernel", 12, ms, S_ * (C * rows * ns * 4 + 3 * rows * ns * 4 + rows * ns * 2 + C * rows * 4),
        "first-layer backward (projection gradient + offset-weight gradient), largest SA shape: %d ch x %d centroids x %d neighbours"
        % (C, rows, ns))
    return out
