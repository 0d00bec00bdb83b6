"""CPU, world_size 2, gloo: the gradient all-reduce path (ratrack_amd/ddp.py) -- bucket construction with
unused parameters, averaging equal to the single-process full-batch gradient, rank-consistent updates."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from ratrack_amd.ddp import FlatGradAllReducer, broadcast_parameters, shard_batch


class Toy(nn.Module):
    """Stand-in with the properties that matter: BN, a used trunk and parameters that never get a gradient."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 16)
        self.bn = nn.BatchNorm1d(16)
        self.b = nn.Linear(16, 3)
        self.dead = nn.Linear(7, 7)                 # never used in forward
        self.bin_score = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        return self.b(torch.relu(self.bn(self.a(x))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)               # ranks start different on purpose
        net = Toy()
        broadcast_parameters(net, 0)
        torch.manual_seed(0)
        full = {"x": torch.randn(8, 6), "y": torch.randn(8, 3)}
        mine = shard_batch(full, rank, world)
        red = FlatGradAllReducer(net)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        net.eval()                                   # eval-mode BN: per-sample independent -> exact comparison possible
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = ((net(mine["x"]) - mine["y"]) ** 2).mean()
            loss.backward()
            red.reduce()
            opt.step()
        # a slot for every trainable parameter (fixed size: every rank always enters the same collective), gradients only for the live ones
        assert "dead.weight" in red.names and "a.weight" in red.names and net.dead.weight.grad is None
        assert red.bucket_bytes == 4 * (sum(p.numel() for p in net.parameters() if p.requires_grad) + red.GUARD)
        # plain numpy: torch tensors travel through shared-memory handles that die with the worker
        q.put((rank, {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}, red.payload_bytes))
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, sd, payload = q.get(timeout=120)
        res[r] = (sd, payload)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # ranks agree bit for bit
    import numpy as np
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    assert res[0][1] == 4 * (6 * 16 + 16 + 16 + 16 + 16 * 3 + 3)     # live parameters only
    # and equal the single-process run on the full batch (mean loss over equal shards == mean over the batch)
    torch.manual_seed(100)
    net = Toy()
    torch.manual_seed(0)
    full = {"x": torch.randn(8, 6), "y": torch.randn(8, 3)}
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    net.eval()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        ((net(full["x"]) - full["y"]) ** 2).mean().backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert np.allclose(v.numpy(), res[0][0][k], rtol=1e-5, atol=1e-6), k


def test_single_process_reducer_is_identity():
    net = Toy()
    x, y = torch.randn(4, 6), torch.randn(4, 3)
    ((net(x) - y) ** 2).mean().backward()
    before = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    red = FlatGradAllReducer(net)
    red.reduce()
    for n, p in net.named_parameters():
        if n in before:
            assert torch.equal(p.grad, before[n])
        else:
            assert p.grad is None


# ---- the Trainer itself (ratrack_amd/train.py) over gloo --------------------------------------------------------------

class TinyBackbone(nn.Module):
    """CPU stand-in exposing Track4D.backbone()'s signature and 7-tuple: a segmentation branch, a flow branch (no gradient
    under the pre-training loss, main_utils.py:148) and parameters that never receive one."""

    def __init__(self):
        super().__init__()
        self.seg = nn.Conv1d(5, 8, 1)
        self.bn = nn.BatchNorm1d(8)
        self.seg_out = nn.Conv1d(8, 1, 1)
        self.flow = nn.Conv1d(5, 3, 1)
        self.dead = nn.Linear(4, 4)

    def backbone(self, pc1, pc2, feature1, feature2, h):
        x = torch.cat([pc1, feature1], 1)
        cls = torch.sigmoid(self.seg_out(torch.relu(self.bn(self.seg(x))))).squeeze(1)
        return 0.1 * self.flow(x), h, cls, None, None, None, None


def _tiny_batches(n_steps, b):
    g = torch.Generator().manual_seed(5)
    out = []
    for _ in range(n_steps):
        pc1 = torch.randn(b, 3, 12, generator=g)
        out.append({"pc1": pc1, "pc2": torch.randn(b, 3, 12, generator=g), "feature1": torch.randn(b, 2, 12, generator=g),
                    "feature2": torch.randn(b, 2, 12, generator=g), "gt_warp": pc1 + 0.1 * torch.randn(b, 3, 12, generator=g),
                    "gt_cls": torch.rand(b, 12, generator=g) > 0.5})
    return out


PRETRAIN = [True, True, False, False, True, False]       # bucket grows at step 3; the flow branch loses its gradient at step 5


def _trainer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ratrack_amd.train import Trainer
        torch.manual_seed(200 + rank)
        net = TinyBackbone()
        broadcast_parameters(net, 0)
        tr = Trainer(net, lr=1e-2)
        payloads = []
        for full, pre in zip(_tiny_batches(len(PRETRAIN), 4), PRETRAIN):
            t = shard_batch(full, rank, world)
            tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], torch.zeros(5, 2, 4), pretrain=pre)
            payloads.append(tr.reducer.payload_bytes)
            assert (net.flow.weight.grad is None) == pre          # no gradient under the pre-training loss: Adam skips it
            assert net.dead.weight.grad is None
        assert "flow.weight" in tr.reducer.names and "dead.weight" in tr.reducer.names
        q.put((rank, {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}, payloads))
    finally:
        dist.destroy_process_group()


def _diverging_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ratrack_amd.train import Trainer
        torch.manual_seed(300)
        net = TinyBackbone()
        tr = Trainer(net, lr=1e-2)
        t = shard_batch(next(iter(_tiny_batches(1, 4))), rank, world)
        try:      # rank 1 runs the pre-training loss, rank 0 the full loss: different live sets
            tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], torch.zeros(5, 2, 4), pretrain=(rank == 1))
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_diverging_live_sets_raise_instead_of_hanging():
    """Advisor r2: ranks whose sets of parameters with a gradient differ used to enter DIFFERENT collectives (a hang).  The bucket
    has a fixed size now and the live-set digest rides in the same all-reduce: both ranks get the error, before the optimizer step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diverging_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all("gradient buckets differ across ranks" in m for m in msgs.values()), msgs


def test_trainer_over_gloo_matches_manual_gradient_averaging():
    """Trainer + FlatGradAllReducer with world_size 2: identical replicas after every loss configuration (incl. a bucket that
    GROWS when the flow branch starts receiving gradients and a parameter that LOSES its gradient again), equal to a
    single process that averages the two shards' gradients by hand.  BatchNorm statistics stay per replica, as under the
    reference's nn.DataParallel: every rank sees its own shard."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, sd, payloads = q.get(timeout=120)
        res[r] = (sd, payloads)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seg_bytes = 4 * (5 * 8 + 8 + 8 + 8 + 8 + 1)
    full_bytes = seg_bytes + 4 * 18
    assert res[0][1] == [seg_bytes, seg_bytes, full_bytes, full_bytes, seg_bytes, full_bytes], res[0][1]      # live gradients of each step
    for k in res[0][0]:
        if "running" in k:
            continue                                              # per-replica BatchNorm statistics
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    # manual reference: two replicas in one process, gradients averaged by hand, one Adam
    from ratrack_amd import loss as L
    from ratrack_amd.train import make_optimizer
    torch.manual_seed(200)
    nets = [TinyBackbone(), TinyBackbone()]
    nets[1].load_state_dict(nets[0].state_dict())
    opt, _ = make_optimizer(nets[0], 1e-2)
    for full, pre in zip(_tiny_batches(len(PRETRAIN), 4), PRETRAIN):
        grads = []
        for r, net in enumerate(nets):
            net.train()
            net.zero_grad(set_to_none=True)
            t = shard_batch(full, r, world)
            flow, _, cls, *_ = net.backbone(t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
            L.backbone_loss(t["pc1"] + flow, cls, t["gt_warp"], t["gt_cls"], pretrain=pre)[0].backward()
            grads.append([p.grad for p in net.parameters()])
        for p, g0, g1 in zip(nets[0].parameters(), *grads):
            p.grad = None if g0 is None else (g0 + g1) / 2
        opt.step()
        with torch.no_grad():
            for p0, p1 in zip(nets[0].parameters(), nets[1].parameters()):
                p1.copy_(p0)
    for k, v in nets[0].state_dict().items():
        if "running" in k or "num_batches" in k:
            continue
        assert np.allclose(v.numpy(), res[0][0][k], rtol=1e-5, atol=1e-6), k
