"""Post-backbone half of Track4D.forward: moving-point clustering and frame-to-frame object association
(reference: models/track4d.py:53-65,108-223; helpers models/utils/track4d_utils.py:19-23,405-438).
SURVEY.md 8(f) ranks 1-2 -- the first consumers of the backbone's cls / flow / prop outputs.  B = 1 logic, as in
the reference (mov_mask = (cls > 0.5).squeeze(0)).

The reference clusters with sklearn.cluster.DBSCAN on the host; `dbscan` below restates that algorithm (same core
definition, same expansion order, hence the same labels -- tests/test_association_cpu.py checks it against sklearn
itself) so the product does not depend on scikit-learn.  Everything else is PyTorch and device agnostic.
"""
from collections import defaultdict

import numpy as np
import torch


def dbscan(x, eps, min_samples):
    """Labels of sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(x) (Euclidean, brute force): x (n, d) array.
    A point is core when its closed eps-ball holds >= min_samples points (itself included); clusters grow by a
    depth-first expansion from the core points in index order; -1 = noise."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    labels = np.full(n, -1, dtype=np.int64)
    if n == 0:
        return labels
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    neigh = [np.nonzero(np.sqrt(d2[i]) <= eps)[0] for i in range(n)]
    core = np.array([len(v) >= min_samples for v in neigh])
    label = 0
    for i in range(n):
        if labels[i] != -1 or not core[i]:
            continue
        stack, cur = [], i
        while True:
            if labels[cur] == -1:
                labels[cur] = label
                if core[cur]:
                    for v in neigh[cur]:
                        if labels[v] == -1:
                            stack.append(v)
            if not stack:
                break
            cur = stack.pop()
        label += 1
    return labels


def obj_centre(obj):
    """(1,3,n) -> (1,3) mean position (track4d_utils.py:19-23)."""
    return torch.mean(obj, dim=2)


def cluster_objects(point_features, eps=1.5, min_samples=2):
    """models/track4d.py:108-126.  point_features (1,139,n) = [warped xyz | xyz | flow | (RCS, v_r) | prop(128)] of the
    moving points; clustered on channels 3:9 and 10:12 (xyz, flow, v_r, first prop channel).  Returns the objects as a
    list of (1,139,n_i) tensors, in order of their first point."""
    if point_features.shape[2] == 0:
        return []
    f = torch.cat((point_features[0, 3:9, :], point_features[0, 10:12, :]), dim=0).detach().cpu().numpy().T
    labels = dbscan(f, eps, min_samples)
    groups = defaultdict(list)
    for i, l in enumerate(labels):
        if l != -1:
            groups[int(l)].append(i)
    if not groups:
        return []
    # one gather for all objects (one host->device index copy), then per-object views of it
    order = [i for ix in groups.values() for i in ix]
    gathered = point_features.index_select(2, torch.as_tensor(order, device=point_features.device))
    return list(torch.split(gathered, [len(ix) for ix in groups.values()], dim=2))


_CHANNELS = {}


def cluster_objects_device(point_features, cls, eps=1.5, min_samples=2, threshold=0.5):
    """cluster_objects(point_features[:, :, cls > threshold]) without the boolean-mask gather, the feature download and the
    host-side clustering: ONE kernel (rtk_dbscan: mover selection + DBSCAN, labels identical to `dbscan`) and ONE small
    device->host copy of the per-point labels (the object list is host-structured: a dict of per-object tensors).
    point_features (1,139,N) CUDA fp32, cls (1,N) motion-segmentation probabilities."""
    from . import _lib, fused  # noqa: F401  (fused registers the signatures)
    pf = point_features.contiguous()
    n = pf.shape[2]
    dev = pf.device
    chan = _CHANNELS.get(dev)
    if chan is None:
        chan = _CHANNELS[dev] = torch.tensor([3, 4, 5, 6, 7, 8, 10, 11], dtype=torch.int32, device=dev)
    score = cls.detach().reshape(-1).contiguous().float()
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.call("rtk_dbscan", n, pf.detach().data_ptr(), n, chan.data_ptr(), score.data_ptr(), float(threshold), float(eps), int(min_samples),
              labels.data_ptr(), torch.cuda.current_stream().cuda_stream)
    lab = labels.cpu().numpy()
    pts = np.nonzero(lab >= 0)[0]
    if pts.size == 0:
        return []
    # the reference collects the objects in a dict while scanning the points in index order (models/track4d.py:119-125):
    # objects are ordered by their FIRST MEMBER POINT (= cluster-id order unless a border point precedes its cluster's cores)
    ids, first = np.unique(lab[pts], return_index=True)
    rank = np.empty(int(ids.max()) + 1, dtype=np.int64)
    rank[ids[np.argsort(first, kind="stable")]] = np.arange(ids.size)
    key = rank[lab[pts]]
    order = pts[np.argsort(key, kind="stable")]               # objects in reference order, points in index order inside
    sizes = np.bincount(key, minlength=ids.size).tolist()
    gathered = pf.index_select(2, torch.from_numpy(order).to(dev))
    return list(torch.split(gathered, sizes, dim=2))


def object_descriptor(obj, prop_channels):
    """141-d descriptor of an object (models/track4d.py:200-214): [centre(3) | var xyz(3) | max prop(128) | mean flow(3) |
    mean (RCS,v_r)(2) | var (RCS,v_r)(2)], shape (1,1,141)."""
    feat = torch.max(obj[:, 11:11 + prop_channels, :], dim=2)[0].unsqueeze(1)
    flow = torch.mean(obj[:, 6:9, :], dim=2).unsqueeze(1)
    pos = obj_centre(obj[:, 3:6, :]).unsqueeze(1)
    rrv = torch.mean(obj[:, 9:11, :], dim=2).unsqueeze(1)
    rrv_var = torch.var(obj[:, 9:11, :], dim=2, unbiased=False).unsqueeze(1)
    var = torch.var(obj[:, 3:6, :], dim=2, unbiased=False).unsqueeze(1)
    return torch.cat((pos, var, feat, flow, rrv, rrv_var), dim=2)


def batched_descriptors(objs):
    """object_descriptor(o, 128) of every object in `objs` ((1,139,n_i) tensors) with a dozen launches in total instead of a
    dozen per object: the objects' points are concatenated and reduced per segment (two-pass variance, like torch.var).
    Returns (len(objs), 141)."""
    dev = objs[0].device
    x = torch.cat([o[0] for o in objs], dim=1)                                         # (139, P)
    sizes = [o.shape[2] for o in objs]
    seg = torch.repeat_interleave(torch.arange(len(objs), device=dev), torch.tensor(sizes, device=dev), output_size=sum(sizes))
    cnt = torch.tensor(sizes, device=dev, dtype=x.dtype)
    k = len(objs)
    stat = x[3:11]                                                                     # xyz(3) | flow(3) | (RCS, v_r)(2)
    mean = torch.zeros(8, k, device=dev, dtype=x.dtype).index_add_(1, seg, stat) / cnt
    dev2 = (stat - mean[:, seg]) ** 2
    var = torch.zeros(8, k, device=dev, dtype=x.dtype).index_add_(1, seg, dev2) / cnt
    prop = x[11:139]
    feat = torch.full((128, k), float("-inf"), device=dev, dtype=x.dtype).scatter_reduce_(1, seg.expand(128, -1), prop, "amax")
    # [centre(3) | var xyz(3) | max prop(128) | mean flow(3) | mean (RCS,v_r)(2) | var (RCS,v_r)(2)]
    return torch.cat((mean[0:3], var[0:3], feat, mean[3:6], mean[6:8], var[6:8]), dim=0).t().contiguous()


def affinity_matrix(affinity_net, objects_curr, objects_prev, descriptors=None):
    """M (previous) x N (current) affinities (models/track4d.py:182-223).  Returns (aff_list, aff_mat (1,M,N), M, N).
    The reference evaluates the descriptor pair and the 5-layer Affinity MLP once per (previous, current) pair -- M*N
    tiny launches sequences, 60-100 ms per frame for 20 x 20 objects.  Here every object's descriptor is computed once
    (`descriptors`: optional cache keyed by id(object tensor), filled here -- an object keeps its descriptor when it
    becomes a previous object in the next frame: the reference's 11:267 slice of a 139-channel tensor IS its 128 prop
    channels) and the MLP runs once on the (M*N, 141) matrix of descriptor differences, in the reference's pair order."""
    m, n = len(objects_prev), len(objects_curr)
    keys = list(objects_prev.keys())
    dev = objects_curr[0].device if n else (objects_prev[keys[0]].device if m else torch.device("cpu"))
    if m == 0 or n == 0:        # nothing to associate (the reference ends up with an empty tensor and starts new tracks)
        return [], torch.zeros(1, m, n, device=dev), m, n
    # the cross-frame cache holds tensors: under autograd they would drag the previous frame's (freed) graph into this
    # frame's affinity loss, so training recomputes every descriptor from the tensors it is handed, as the reference does
    cache = descriptors if (descriptors is not None and not torch.is_grad_enabled()) else {}

    todo, seen = [], set()
    for o in list(objects_curr) + [objects_prev[k] for k in keys]:
        if id(o) not in cache and id(o) not in seen:
            seen.add(id(o))
            todo.append(o)
    if todo:
        for o, d in zip(todo, batched_descriptors(todo)):
            cache[id(o)] = d.view(1, 1, -1)

    def desc(obj):
        return cache[id(obj)]

    d_curr = torch.cat([desc(objects_curr[j]) for j in range(n)], dim=1)           # (1,N,141)
    d_prev = torch.cat([desc(objects_prev[k]) for k in keys], dim=1)               # (1,M,141)
    diff = (d_curr.unsqueeze(1) - d_prev.unsqueeze(2)).reshape(m * n, -1)          # pair (i, j) -> row i*N + j: curr_j - prev_i
    aff_list = affinity_net.affinity(diff).reshape(m * n)
    return aff_list, aff_list.reshape(m, n).unsqueeze(0), m, n


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    """track4d_utils.py:405-411."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters):
    """Differentiable optimal transport in log space with a dustbin row/column (track4d_utils.py:414-434)."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def _log_optimal_transport_hip(aff_mat, alpha, iters):
    """log_optimal_transport on the GPU as ONE kernel (rtk_log_sinkhorn) instead of ~8 framework kernels per iteration."""
    from . import _lib, fused  # noqa: F401  (fused registers the signature)
    _, m, n = aff_mat.shape
    scores = aff_mat.detach().reshape(m, n).contiguous().float()
    out = torch.empty(1, m + 1, n + 1, dtype=torch.float32, device=aff_mat.device)
    _lib.call("rtk_log_sinkhorn", m, n, scores.data_ptr(), float(alpha), int(iters), out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return out


def sinkhorn_assignment(aff_mat, iters=500):
    """Mutual-best assignment after Sinkhorn normalisation (models/track4d.py:166-180): for every current object the
    index of the matched previous object, or -1.  Returns indices1 (1,N) int64."""
    _, m, n = aff_mat.shape
    if aff_mat.is_cuda and (m + 1) * ((n + 1) | 1) + m + n + 2 <= 16 * 1024:
        scores = _log_optimal_transport_hip(aff_mat, 0.9, iters)
    else:       # host tensors (CPU tests) or more objects than one workgroup's LDS holds: the framework formulation
        scores = log_optimal_transport(aff_mat, torch.tensor(0.9, device=aff_mat.device), iters)
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    indices0, indices1 = max0.indices, max1.indices
    ar = lambda t: t.new_ones(t.shape[1]).cumsum(0) - 1
    mutual0 = ar(indices0)[None] == indices1.gather(1, indices0)
    mutual1 = ar(indices1)[None] == indices0.gather(1, indices1)
    # python scalars, not new_tensor(): each new_tensor is a synchronous host->device copy (~1 ms)
    mscores0 = torch.where(mutual0, max0.values.exp(), 0.0)
    valid0 = mutual0 & (mscores0 > 0)
    valid1 = mutual1 & valid0.gather(1, indices1)
    return torch.where(valid1, indices1, -1)


class Associator:
    """ID bookkeeping across frames (models/track4d.py:135-164): holds `max_id`."""

    def __init__(self, affinity_net):
        self.affinity_net = affinity_net
        self.max_id = 0
        self._desc = {}          # id(object tensor) -> its 141-d descriptor, for the objects of the last frame

    def __call__(self, objects_curr, objects_prev):
        objects, confs = dict(), []
        live = {id(o): o for o in list(objects_prev.values()) + list(objects_curr)}
        self._desc = {k: v for k, v in self._desc.items() if k in live}      # descriptors of objects that still exist
        aff_list, aff_mat, m, n = affinity_matrix(self.affinity_net, objects_curr, objects_prev, self._desc)
        self._keep = live        # the cached ids stay valid only while the tensors are alive
        indices1 = None

        def fresh(obj):
            objects[self.max_id] = obj
            self.max_id += 1
            confs.append(0)

        if aff_mat.size(1) > 0 and aff_mat.size(2) > 0:
            try:
                indices1 = sinkhorn_assignment(aff_mat)
                prev_keys = list(objects_prev.keys())
                idx_host = indices1[0].tolist()                       # ONE device->host transfer each for the decisions
                aff_host = aff_mat[0].detach().cpu()
                for i in range(n):
                    k = idx_host[i]
                    # the reference indexes aff_mat[0, -1, i] when k == -1 (python wrap-around) before testing k
                    if k == -1 or k >= m or float(aff_host[k, i]) < 0.01:
                        fresh(objects_curr[i])
                    else:
                        objects[prev_keys[k]] = objects_curr[i]
                        confs.append(aff_mat[0, k, i])
            except Exception:      # the reference swallows association failures and starts new tracks (track4d.py:154-158)
                for obj in objects_curr:
                    fresh(obj)
        else:
            for obj in objects_curr:
                fresh(obj)
        return aff_list, aff_mat, indices1, confs, objects
