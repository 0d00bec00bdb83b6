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
        assert "dead.weight" not in red.names and "bin_score" not in red.names and "a.weight" in red.names
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
