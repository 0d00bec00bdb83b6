"""CPU: the post-backbone half of forward() (ratrack_amd/association.py) -- DBSCAN restatement against scikit-learn,
and clustering + affinity + Sinkhorn association + ID bookkeeping against the golden vectors captured from the
reference's own forward() over two consecutive frames (tools/make_golden.py: forward_case)."""
import numpy as np
import pytest
import torch

from ratrack_amd import association as A
from ratrack_amd.track4d import Args, Track4D

from _util import load_case, reference_state_dict

CLS_BIAS_SHIFT = 0.09       # tools/make_golden.py FORWARD_CLS_BIAS_SHIFT


@pytest.mark.parametrize("n,d,eps,ms,seed", [(240, 8, 1.5, 2, 0), (240, 8, 1.5, 4, 1), (100, 3, 0.3, 3, 2), (60, 2, 0.2, 5, 3), (1, 8, 1.5, 2, 4)])
def test_dbscan_matches_sklearn(n, d, eps, ms, seed):
    sk = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 3, (6, d))
    x = (centres[rng.integers(0, 6, n)] + rng.normal(0, 0.4, (n, d))).astype(np.float32)
    x[::7] = rng.normal(0, 8, (len(x[::7]), d))           # outliers -> noise / border points
    ref = sk.DBSCAN(eps=eps, min_samples=ms).fit_predict(x)
    assert np.array_equal(A.dbscan(x, eps, ms), ref)


def test_forward_second_half_matches_reference():
    from oracle import track4d_ref as R
    case = load_case("forward_b1_n256")
    sd = reference_state_dict()
    sd["fd_layer.cp.linear.bias"] = sd["fd_layer.cp.linear.bias"] + CLS_BIAS_SHIFT
    net = Track4D(Args())                                     # CPU instance: only its Affinity MLP + bookkeeping are used
    net.load_state_dict(sd, strict=True)
    net.eval()
    objects_prev, h = dict(), torch.zeros(5, 1, 128)
    with torch.no_grad():
        for fi in range(2):
            g = lambda k: torch.from_numpy(case["f%d_in_%s" % (fi, k)])
            pc1, pc2, f1, f2 = g("pc1"), g("pc2"), g("feature1"), g("feature2")
            flow, h, cls, cor, pf1, pf2, prop = R.backbone(sd, pc1, pc2, f1, f2, h)      # oracle backbone == reference (pinned elsewhere)
            p = "f%d_" % fi
            assert np.abs(cls.numpy() - case[p + "cls"]).max() < 1e-5
            pc1_warp, aff_list, aff_mat, indices1, confs, objects, objects_curr = net.detect_and_associate(pc1, f1, flow, cls, prop, objects_prev)
            assert np.abs(pc1_warp.numpy() - case[p + "pc1_warp"]).max() < 1e-5
            assert len(objects_curr) == int(case[p + "n_objects_curr"])
            assert [o.shape[2] for o in objects_curr] == case[p + "object_sizes_curr"].tolist()
            first = np.array([o[0, 3:6, 0].numpy() for o in objects_curr]).reshape(-1, 3)
            assert np.allclose(first, case[p + "object_first_xyz"], atol=1e-6)
            assert list(objects.keys()) == case[p + "object_ids"].tolist()
            assert [objects[k].shape[2] for k in objects] == case[p + "object_sizes"].tolist()
            assert net.max_id == int(case[p + "max_id"])
            ref_aff = case[p + "aff_mat"]
            if ref_aff.size:
                assert aff_mat.shape == ref_aff.shape and np.abs(aff_mat.numpy() - ref_aff).max() < 2e-5
                assert np.array_equal(indices1.numpy(), case[p + "indices1"])
            assert np.allclose([float(c) for c in confs], case[p + "confs"], atol=2e-5)
            objects_prev = {k: v.clone().detach() for k, v in objects.items()}
    assert len(case["f1_confs"]) and (case["f1_confs"] > 0).any()      # the fixture really exercises re-identification


def test_batched_affinity_matches_per_pair_evaluation():
    """affinity_matrix evaluates the Affinity MLP once on all (previous, current) pairs; the reference evaluates it pair by
    pair (models/track4d.py:182-223).  Same values, same aff_list order, descriptors cached by object identity."""
    import torch
    from ratrack_amd import association as A
    from ratrack_amd.track4d import Affinity
    torch.manual_seed(0)
    net = Affinity(141)
    curr = [torch.randn(1, 139, k) for k in (3, 1, 7, 2)]
    prev = {5: torch.randn(1, 139, 4), 9: torch.randn(1, 139, 2), 11: curr[2]}
    cache = {}
    aff_list, aff_mat, m, n = A.affinity_matrix(net, curr, prev, cache)
    assert (m, n) == (3, 4) and aff_mat.shape == (1, 3, 4) and aff_list.shape == (12,)
    want = []
    for key in prev:                                                    # the reference's loop nest and slices
        for o2 in curr:
            d2 = A.object_descriptor(o2, 128)
            d1 = A.object_descriptor(prev[key], 256)
            want.append(net(d2, d1).squeeze(0))
    want = torch.cat(want)
    assert torch.allclose(aff_list, want, rtol=1e-5, atol=1e-6)
    assert torch.allclose(aff_mat[0], want.reshape(3, 4), rtol=1e-5, atol=1e-6)
    # under autograd the cross-frame cache is bypassed (a cached descriptor would drag the previous frame's freed graph into this
    # frame's affinity loss); in inference it holds 4 current + 2 distinct previous objects
    assert len(cache) == 0
    with torch.no_grad():
        aff_nograd, _, _, _ = A.affinity_matrix(net, curr, prev, cache)
    assert len(cache) == 6 and id(curr[2]) in cache and torch.allclose(aff_nograd, want, rtol=1e-5, atol=1e-6)
    # empty sides
    l0, m0, a, b = A.affinity_matrix(net, [], prev)
    assert m0.shape == (1, 3, 0) and (a, b) == (3, 0)


def test_batched_descriptors_match_per_object_descriptor():
    import torch
    from ratrack_amd import association as A
    torch.manual_seed(1)
    objs = [torch.randn(1, 139, k) for k in (1, 5, 2, 9, 3)]
    got = A.batched_descriptors(objs)
    want = torch.cat([A.object_descriptor(o, 128).reshape(1, 141) for o in objs], 0)
    assert got.shape == (5, 141)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
