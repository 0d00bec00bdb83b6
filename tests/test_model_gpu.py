"""GPU: the product Track4D.backbone (module path: HIP ops + PyTorch-ROCm dense layers) against the
golden vectors captured from the reference graph and against the CPU oracle."""
import numpy as np
import pytest
import torch

from ratrack_amd.track4d import Args, Track4D

from _util import EVAL_CASES, RTOL, assert_close, inputs_of, load_case, reference_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_net(train=False):
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(reference_state_dict(DEV), strict=True)
    net.train(train)
    return net


@pytest.mark.parametrize("name", EVAL_CASES)
def test_backbone_modules_match_golden(name):
    case = load_case(name)
    net = make_net()
    net._use_fused = False
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    with torch.no_grad():
        flow, h, cls, cor, pf1, pf2, prop = net.backbone(pc1, pc2, f1, f2, None)
        flow2, h2, *_ = net.backbone(pc1, pc2, f1, f2, h)
    cpu = lambda t: t.float().cpu().numpy()
    assert_close(cpu(flow), case["flow"], RTOL, "flow")
    assert_close(cpu(cls), case["cls"], RTOL, "cls")
    assert_close(cpu(h), case["h_out"], RTOL, "h")
    assert_close(cpu(cor[:, :, ::8]), case["cor_s8"], RTOL, "cor")
    assert_close(cpu(prop[:, :, ::8]), case["prop_s8"], RTOL, "prop")
    assert_close(cpu(pf1[:, :, ::8]), case["pc1_features_s8"], RTOL, "pc1_features")
    assert_close(cpu(pf2[:, :, ::8]), case["pc2_features_s8"], RTOL, "pc2_features")
    assert_close(cpu(flow2), case["flow_step2"], RTOL, "flow step 2")
    assert_close(cpu(h2), case["h_out_step2"], RTOL, "h step 2")
    # scene-flow EPE vs the reference's (headline metric's second half)
    gt = torch.from_numpy(case["in_gt_warp"]).to(DEV)
    epe = float(torch.sqrt(((pc1[:1] + flow[:1] - gt[:1]) ** 2).sum(1) + 1e-20).mean())
    ref = float(case["metric_sf_vals"][list(case["metric_sf_keys"]).index("epe")])
    assert abs(epe - ref) <= 1e-4 * max(ref, 1.0), (epe, ref)


def test_native_indices_match_golden():
    """FPS / ball-query / three_nn / kNN index tensors of the first PNHead call, bit-exact."""
    from ratrack_amd import pointnet2_utils as PU
    for name in EVAL_CASES:
        case = load_case(name)
        pc1, pc2, _, _ = inputs_of(case, DEV)
        xyz = pc1.permute(0, 2, 1).contiguous()
        radii = [case["ball_radius_%d" % i] for i in range(6)]
        ns = [4, 8, 8, 16, 16, 32]
        cur = xyz
        levels = []
        for lvl in range(3):
            idx = PU.furthest_point_sample(cur, 512)
            assert np.array_equal(idx.cpu().numpy(), case["fps_idx_c0_l%d" % (lvl + 1)]), (name, lvl)
            new_xyz = PU.gather_operation(cur.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            for s in range(2):
                i = lvl * 2 + s
                b = PU.ball_query(float(radii[i]), ns[i], cur, new_xyz)
                assert np.array_equal(b.cpu().numpy(), case["ball_idx_%d" % i]), (name, i)
            levels.append((cur, new_xyz))
            cur = new_xyz
        l0, l1, l2, l3 = xyz, levels[0][1], levels[1][1], levels[2][1]
        for i, (u, k) in enumerate([(l2, l3), (l1, l2), (l0, l1)]):
            dist, idx = PU.three_nn(u, k)
            assert np.array_equal(idx.cpu().numpy(), case["three_nn_idx_%d" % i]), (name, i)
            assert np.array_equal((dist * dist).cpu().numpy(), case["three_nn_dist2_%d" % i]) or \
                np.allclose((dist * dist).cpu().numpy(), case["three_nn_dist2_%d" % i], rtol=1e-6, atol=0)
        p1 = pc1.permute(0, 2, 1).contiguous()
        p2 = pc2.permute(0, 2, 1).contiguous()
        for i, (src, q) in enumerate([(p2, p1), (p1, p1)]):
            k = np.sort(PU.knn_point(16, src, q).cpu().numpy(), axis=-1)
            ok = case["knn_kth_gap_%d" % i] > 0
            assert np.array_equal(k[ok], case["knn_set_%d" % i][ok]), (name, i)


def test_train_step_matches_golden():
    """B=1 train-mode forward + multi-task loss + backward on the GPU module path."""
    from ratrack_amd import loss as L
    case = load_case("train_b1_n256")
    net = make_net(train=True)
    pc1, pc2, f1, f2 = inputs_of(case, DEV)
    flow, h, cls, *_ = net.backbone(pc1, pc2, f1, f2, None)
    gt = torch.from_numpy(case["in_gt_warp"]).to(DEV)
    gt_cls = torch.from_numpy(case["in_gt_cls"]).to(DEV)
    total, items = L.backbone_loss(pc1 + flow, cls, gt, gt_cls, pretrain=False)
    keys = [str(k) for k in case["loss_keys"]]
    np.testing.assert_allclose([float(items[k]) for k in keys], case["loss_vals"], rtol=1e-4, atol=1e-6)
    total.backward()
    names = [str(k) for k in case["grad_names"]]
    params = dict(net.named_parameters())
    gmax = float(case["grad_norms"].max())
    for k, ref in zip(names, case["grad_norms"]):
        g = params[k].grad
        if ref < 0:
            assert g is None or float(g.norm()) == 0.0, k
        else:
            assert abs(float(g.norm()) - ref) <= 5e-3 * ref + 1e-5 * gmax, (k, float(g.norm()), ref)
    sd = net.state_dict()
    for k in case:
        if k.startswith("bn/") and not k.endswith("num_batches_tracked"):
            assert_close(sd[k[3:]].cpu().numpy(), case[k], 1e-4, k)


def test_full_forward_matches_reference_two_frames():
    """Track4D.forward (fused backbone + clustering + association) on the GPU over two consecutive frames against the
    reference's own forward(): movers, objects, affinities, assignments, track IDs."""
    case = load_case("forward_b1_n256")
    sd = reference_state_dict(DEV)
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09          # tools/make_golden.py FORWARD_CLS_BIAS_SHIFT
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(sd, strict=True)
    net.eval()
    objects_prev, h = dict(), torch.zeros(5, 1, 128, device=DEV)
    with torch.no_grad():
        for fi in range(2):
            g = lambda k: torch.from_numpy(case["f%d_in_%s" % (fi, k)]).to(DEV)
            h, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, timeout, objects_curr = net(
                g("pc1"), g("pc2"), g("feature1"), g("feature2"), h, objects_prev)
            p = "f%d_" % fi
            assert_close(pc1_warp.cpu().numpy(), case[p + "pc1_warp"], RTOL, "pc1_warp")
            assert np.abs(cls.cpu().numpy() - case[p + "cls"]).max() < 1e-5
            assert [o.shape[2] for o in objects_curr] == case[p + "object_sizes_curr"].tolist()
            assert list(objects.keys()) == case[p + "object_ids"].tolist()
            if case[p + "aff_mat"].size:
                assert np.abs(aff_mat.cpu().numpy() - case[p + "aff_mat"]).max() < 1e-4
                assert np.array_equal(indices1.cpu().numpy(), case[p + "indices1"])
            assert np.allclose([float(c) for c in confs], case[p + "confs"], atol=1e-4)
            objects_prev = {k: v.clone().detach() for k, v in objects.items()}


def test_full_loss_train_step_matches_reference():
    """ONE TRAINING ITERATION AFTER PRE-TRAINING, as the reference's epoch loop runs it (main_utils.py:127-156): net.train(), frame 0
    through forward() (its objects, detached, become objects_prev), frame 1 through forward() with them, total = 0.5 L_sf + 0.5 L_trk +
    L_seg through the 19-argument track_4d_loss (losses/loss.py:8-31,48-72), backward.  Fixture: the same iteration through the
    imported reference (tools/make_golden.py full_loss_case): the three loss terms, the Affinity MLP's list, and EVERY parameter's
    gradient tensor -- the tracking term's gradient reaches affinity.* directly and the backbone through the pooled object
    descriptors (fd_layer.* / pn_head.*)."""
    from _util import grad_sample, probe_vector
    from ratrack_amd import loss as L
    case = load_case("train_full_b1_n256")
    sd = reference_state_dict(DEV)
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09          # tools/make_golden.py FORWARD_CLS_BIAS_SHIFT
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(sd, strict=True)
    net.train()
    g = lambda fi, k: torch.from_numpy(case["f%d_in_%s" % (fi, k)]).to(DEV)
    h = torch.zeros(5, 1, 128, device=DEV)
    h, _, _, _, _, _, _, objects, _, _ = net(g(0, "pc1"), g(0, "pc2"), g(0, "feature1"), g(0, "feature2"), h, dict())
    assert list(objects.keys()) == case["f0_object_ids"].tolist()
    assert [objects[k].shape[2] for k in objects] == case["f0_object_sizes"].tolist()
    objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    net.zero_grad()
    h1, pc1_warp, cls, aff_list, aff_mat, indices1, confs, objects, _, objects_curr = net(
        g(1, "pc1"), g(1, "pc2"), g(1, "feature1"), g(1, "feature2"), h.detach(), objects_prev)
    assert [o.shape[2] for o in objects_curr] == case["f1_object_sizes_curr"].tolist()
    assert_close(pc1_warp.detach().cpu().numpy(), case["f1_pc1_warp"], RTOL, "pc1_warp (train mode)")
    assert np.abs(cls.detach().cpu().numpy() - case["f1_cls"]).max() < 1e-5
    assert np.abs(aff_list.detach().cpu().numpy().reshape(-1) - case["aff_list"].reshape(-1)).max() < 1e-4
    mp = {int(k): i for i, k in enumerate(case["prev_keys"])}
    mc = {int(k): i for i, k in enumerate(case["curr_keys"])}
    gt, gt_cls = g(1, "gt_warp"), torch.from_numpy(case["f1_gt_cls_used"]).to(DEV)
    total, items = L.track_4d_loss(objects_prev, objects, mp, mc, None, None, None, g(1, "pc1"), g(1, "pc2"), pc1_warp, cls, gt, aff_list,
                                   None, gt_cls, None, None, None, pretrain=False)
    keys = [str(k) for k in case["loss_keys"]]
    np.testing.assert_allclose([float(items[k]) for k in keys], case["loss_vals"], rtol=1e-4, atol=1e-6)
    assert float(items["TrackingLoss"]) > 0.1
    total.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().float().cpu().numpy()) for k, p in net.named_parameters()}
    # per tensor: max|mine - reference| over the fixture's sampled elements / the reference tensor's largest element, and the
    # whole-tensor probe product.  Parameters whose exact gradient is zero (a bias in front of a BatchNorm, the GRU under B = 1 batch
    # statistics) carry rounding noise in the reference as well: below 1e-5 of the model's largest gradient element they only have
    # to be noise here too.
    names = [str(k) for k in case["grad_names"]]
    gmax = max(float(np.abs(case["grad/" + k]).max()) for k, n in zip(names, case["grad_norms"]) if n >= 0)
    live = []
    for i, (k, n) in enumerate(zip(names, case["grad_norms"])):
        gk = grads.get(k)
        if n < 0:
            assert gk is None or float(np.abs(gk).max()) == 0.0, "%s: dead parameter has a gradient" % k
            continue
        assert gk is not None, "%s: no gradient" % k
        ref, mine = case["grad/" + k].astype(np.float64), grad_sample(gk)
        if float(np.abs(ref).max()) <= 1e-5 * gmax:
            assert float(np.abs(mine).max()) <= 1e-4 * gmax, (k, float(np.abs(mine).max()), gmax)
            continue
        full = np.asarray(gk, dtype=np.float64).ravel()
        live.append(dict(name=k, e_ref=float(np.abs(mine - ref).max() / np.abs(ref).max()), probe=float((full * probe_vector(k, full.size)).sum()),
                         ref_probe=float(case["grad_probes"][i]), ref_norm=float(n)))
    assert len(live) > 150
    for prefix in ("affinity.", "fd_layer.cp.", "fd_layer.", "pn_head."):
        assert any(r["name"].startswith(prefix) for r in live), prefix
    worst = sorted(live, key=lambda r: -r["e_ref"])[:5]
    print("\nfull-loss train step: worst gradient tensors vs the reference: " + ", ".join("%s %.1e" % (r["name"], r["e_ref"]) for r in worst))
    # per tensor max|a - b| / max|b| over the sampled elements.  The fixture's frame pairs were chosen free of flipped ReLU / max-pool
    # decisions (tools/make_golden.py full_loss_case), so the bounds are those of the flip-free real pairs of tests/test_varn_train_gpu.py
    err = np.array([r["e_ref"] for r in live])
    print("   %d tensors: median %.1e, 90th percentile %.1e, max %.1e" % (len(live), np.median(err), np.quantile(err, 0.9), err.max()))
    for r in live:
        assert r["e_ref"] <= 5e-3, (r["name"], r["e_ref"])
        assert abs(r["probe"] - r["ref_probe"]) <= 1e-2 * r["ref_norm"], r
    assert np.median(err) <= 5e-4 and np.quantile(err, 0.9) <= 2e-3, (np.median(err), np.quantile(err, 0.9))
    aff = np.array([r["e_ref"] for r in live if r["name"].startswith("affinity.")])
    assert aff.size >= 4 and aff.max() <= 2e-3, aff


def _example_frame(f):
    import os
    from _util import GOLDEN
    ex = os.path.join(GOLDEN, "vod_example")
    lines = []
    for i, line in enumerate(open(os.path.join(ex, "label_%s.txt" % f)).read().splitlines()):      # detection lines -> tracking format, ids in file order
        t = line.split(" ")
        lines.append(" ".join([t[0], str(i)] + t[2:15]))
    return dict(radar=os.path.join(ex, "radar_%s.bin" % f), radar_calib=os.path.join(ex, "radar_calib_%s.txt" % f),
                lidar_calib=os.path.join(ex, "lidar_calib_%s.txt" % f), pose=os.path.join(ex, "pose_%s.json" % f),
                labels=os.path.join(ex, "label_%s.txt" % f), tracking=lines)


def test_gt_train_iteration_on_shipped_frames_matches_reference():
    """SURVEY 8(f3) end to end: the three radar frames the reference ships -> vod_io (clouds, features, ego-motion compensation) ->
    vod_gt (moving labels, oriented boxes, gt_cls, GT objects, GT warped positions) -> TWO passes of the epoch loop after
    pre-training (main_utils.py:66-156): net.train() forward(), map_gt_objects, the 19-argument track_4d_loss, backward -- the
    second pass with the first one's objects and GT mappings.  Fixture: the same two passes through the imported reference
    (tools/make_golden_gt.py --train; its get_gt_flow_new / map_gt_objects / Track4D / track_4d_loss, the boxes handed over as
    (center, R) because Open3D is absent): GT tensors, cluster ids and sizes, mapping keys, loss items of both passes, every
    parameter's gradient of the second."""
    import random
    from _util import grad_sample, probe_vector
    from ratrack_amd import loss as L, vod_gt
    case = load_case("train_gt_real")
    sd = reference_state_dict(DEV)
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09          # tools/make_golden.py FORWARD_CLS_BIAS_SHIFT
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(sd, strict=True)
    net.train()
    objects_prev, mappings_prev, h = dict(), dict(), torch.zeros(5, 1, 128, device=DEV)
    keys = [str(k) for k in case["loss_keys"]]
    pairs = [("01047", "01201"), ("00549", "01047")]
    for it, (later, earlier) in enumerate(pairs):
        pre = "p%d_" % it
        g = vod_gt.frame_pair_gt(_example_frame(later), _example_frame(earlier), device=DEV)
        # ---- the GT the files give, against the reference's own functions on the same files
        assert torch.equal(g.gt_cls.cpu(), torch.from_numpy(case[pre + "gt_cls"]))
        assert np.abs(g.pc1_compensated.cpu().numpy() - case[pre + "pc1_comp"]).max() <= 2e-5        # positions of up to 130 m: 2 ulp
        assert np.abs(g.gt_flow.cpu().numpy() - case[pre + "gt"]).max() <= 2e-5
        assert list(g.gt_objs.keys()) == case[pre + "gt_obj_ids"].tolist()
        net.zero_grad()
        h, pc1_warp, cls, aff_list, aff_mat, assig, confs, objects, _, objects_curr = net(g.pc1, g.pc2, g.feature1, g.feature2, h, objects_prev)
        assert list(objects.keys()) == case[pre + "object_ids"].tolist()
        assert [objects[k].shape[2] for k in objects] == case[pre + "object_sizes"].tolist()
        assert_close(pc1_warp.detach().cpu().numpy(), case[pre + "pc1_warp"], RTOL, "pc1_warp (train mode, pass %d)" % it)
        assert np.abs(cls.detach().cpu().numpy() - case[pre + "cls"]).max() < 1e-5
        random.seed(100 + it)                                                      # unmatched predictions draw negative keys from `random`
        mappings_curr, mappings_inv = vod_gt.map_gt_objects(g.objs_centre, g.gt_objs, objects)
        assert [float(k) for k in mappings_curr.keys()] == case[pre + "map_keys"].tolist()
        assert list(mappings_curr.values()) == case[pre + "map_vals"].tolist()
        total, items = L.track_4d_loss(objects_prev, objects, mappings_prev, mappings_curr, mappings_inv, g.labels1, g.labels2, g.pc1, g.pc2,
                                       pc1_warp, cls, g.gt_flow, aff_list, g.gt_mov_pts, g.gt_cls, g.gt_objs, g.objs_idx, g.objs_centre, pretrain=False)
        np.testing.assert_allclose([float(items[k]) for k in keys], case[pre + "loss_vals"], rtol=1e-4, atol=1e-6)
        if it == 1:
            assert np.abs(aff_list.detach().cpu().numpy().reshape(-1) - case[pre + "aff_list"].reshape(-1)).max() < 1e-4
            assert float(items["TrackingLoss"]) > 0.1 and float(items["SceneFlowLoss"]) > 1.0
            total.backward()
        objects_prev = {k: v.clone().detach() for k, v in objects.items()}
        mappings_prev = mappings_curr
        h = h.detach()
    grads = {k: (None if p.grad is None else p.grad.detach().float().cpu().numpy()) for k, p in net.named_parameters()}
    names = [str(k) for k in case["grad_names"]]
    gmax = max(float(np.abs(case["grad/" + k]).max()) for k, n in zip(names, case["grad_norms"]) if n >= 0)
    live = []
    for i, (k, n) in enumerate(zip(names, case["grad_norms"])):
        gk = grads.get(k)
        if n < 0:
            assert gk is None or float(np.abs(gk).max()) == 0.0, "%s: dead parameter has a gradient" % k
            continue
        assert gk is not None, "%s: no gradient" % k
        ref, mine = case["grad/" + k].astype(np.float64), grad_sample(gk)
        if float(np.abs(ref).max()) <= 1e-5 * gmax:
            assert float(np.abs(mine).max()) <= 1e-4 * gmax, (k, float(np.abs(mine).max()), gmax)
            continue
        full = np.asarray(gk, dtype=np.float64).ravel()
        live.append(dict(name=k, e_ref=float(np.abs(mine - ref).max() / np.abs(ref).max()), probe=float((full * probe_vector(k, full.size)).sum()),
                         ref_probe=float(case["grad_probes"][i]), ref_norm=float(n)))
    assert len(live) > 150
    err = np.array([r["e_ref"] for r in live])
    worst = sorted(live, key=lambda r: -r["e_ref"])[:5]
    print("\nGT train iteration on the shipped frames: %d gradient tensors vs the reference: median %.1e, 90th percentile %.1e, max %.1e; worst: %s"
          % (len(live), np.median(err), np.quantile(err, 0.9), err.max(), ", ".join("%s %.1e" % (r["name"], r["e_ref"]) for r in worst)))
    for prefix in ("affinity.", "fd_layer.cp.", "fd_layer.", "pn_head."):
        assert any(r["name"].startswith(prefix) for r in live), prefix
    # per tensor max|a - b| / max|b| over the sampled elements (measured: median 1.6e-4, 90th percentile 3.4e-4, max 1.5e-3 -- this pair
    # flips no ReLU / max-pool decision between the two fp32 evaluations; the bounds are those of the other flip-free pairs)
    for r in live:
        assert r["e_ref"] <= 5e-3, (r["name"], r["e_ref"])
        assert abs(r["probe"] - r["ref_probe"]) <= 1e-2 * r["ref_norm"], r
    assert np.median(err) <= 5e-4 and np.quantile(err, 0.9) <= 2e-3, (np.median(err), np.quantile(err, 0.9))


def test_result_file_from_gpu_forward(tmp_path):
    """SURVEY 8(f4) on the GPU path: Track4D.forward()'s `objects` / `confs` ON THE DEVICE -> vod_io.write_track_results -> one text file
    per frame (main_utils.py:165-184), for (a) the two consecutive frames of the reference's own forward() fixture -- the file's track
    ids, line count and per-object point counts are the reference's, the confidences its to 1e-4 -- and (b) an epoch-style pass over
    the radar frames the reference ships (files -> vod_gt.frame_pair_gt -> forward()).  Every line is compared with the reference's own
    string construction applied to the device tensors, and parsed back against them."""
    from ratrack_amd import vod_gt, vod_io

    def reference_lines(objects, confs):                                 # main_utils.py:170-182, literally
        out, idx = [], -1
        for obj_id, obj in objects.items():
            idx += 1
            s = "NA"
            s += " 1"
            s += " -1"
            s += " -1"
            s += " " + str(float(confs[idx]))
            s += " " + str(obj_id)
            for i in range(obj.size(2)):
                s += " " + str(float(obj[0, 3, i]))
                s += " " + str(float(obj[0, 4, i]))
                s += " " + str(float(obj[0, 5, i]))
            out.append(s + "\n")
        return out

    def check_file(path, objects, confs, pc1):
        assert all(o.is_cuda for o in objects.values()), "the objects of a GPU forward() live on the device"
        text = open(path).read()
        assert text == "".join(reference_lines(objects, confs))
        rows = vod_io.read_track_results(path)
        assert [r[0] for r in rows] == list(objects.keys())
        cloud = pc1[0].t().double().cpu().numpy()                        # (N, 3): every written point is a point of the input cloud
        for k, (obj_id, conf, pts) in enumerate(rows):
            obj = objects[obj_id]
            assert abs(conf - float(confs[k])) < 1e-12 and 0.0 <= conf <= 1.0
            assert pts.shape == (obj.shape[2], 3)
            assert np.array_equal(pts, obj[0, 3:6].t().double().cpu().numpy())       # str(float(.)) round-trips a float32 exactly
            assert all((np.abs(cloud - p).max(1) == 0).any() for p in pts)
        return rows

    sd = reference_state_dict(DEV)
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + 0.09          # tools/make_golden.py FORWARD_CLS_BIAS_SHIFT
    net = Track4D(Args()).to(DEV)
    net.load_state_dict(sd, strict=True)
    net.eval()
    # (a) the reference's two-frame forward() fixture
    case = load_case("forward_b1_n256")
    objects_prev, h = dict(), torch.zeros(5, 1, 128, device=DEV)
    with torch.no_grad():
        for fi in range(2):
            g = lambda k: torch.from_numpy(case["f%d_in_%s" % (fi, k)]).to(DEV)
            h, _, _, _, _, _, confs, objects, _, _ = net(g("pc1"), g("pc2"), g("feature1"), g("feature2"), h, objects_prev)
            path = vod_io.write_track_results(str(tmp_path), "fixture_seq", fi, objects, confs)
            assert path.endswith("fixture_seq/%05d.txt" % fi)
            rows = check_file(path, objects, confs, g("pc1"))
            p = "f%d_" % fi
            assert [r[0] for r in rows] == case[p + "object_ids"].tolist()
            assert [r[2].shape[0] for r in rows] == case[p + "object_sizes"].tolist()
            assert np.allclose([r[1] for r in rows], case[p + "confs"], atol=1e-4)
            objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    # (b) the shipped radar frames, as the eval loop walks them (later frame = pc1)
    objects_prev, h = dict(), torch.zeros(5, 1, 128, device=DEV)
    written = 0
    with torch.no_grad():
        for index, (later, earlier) in enumerate([("01047", "01201"), ("00549", "01047")]):
            g = vod_gt.frame_pair_gt(_example_frame(later), _example_frame(earlier), device=DEV)
            h, _, _, _, _, _, confs, objects, _, _ = net(g.pc1, g.pc2, g.feature1, g.feature2, h, objects_prev)
            path = vod_io.write_track_results(str(tmp_path), "delft_example", index, objects, confs)
            written += len(check_file(path, objects, confs, g.pc1))
            objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    assert written > 0, "no object on the shipped frames: the file path was not exercised"
