"""Data parallelism for the RaTrack backbone: one process per GPU, gradients averaged with ONE flat
all-reduce over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference's multi-GPU story is nn.DataParallel (models/model.py:38-41): single process, replicas
rebuilt every step, outputs gathered on GPU 0 -- unusable at its batch size of 1.  Frame-pairs are
independent in the backbone and BatchNorm statistics stay per replica (as under DataParallel), so the
only exchange of a training step is the gradient all-reduce:

  * 42 % of the parameters never receive a gradient on the backbone path.  The bucket
    is built from the parameters that HAVE a gradient after the first backward; the set is a property of
    the graph, identical on every rank, and is verified across ranks once (hash all-reduce) -- there is no
    find_unused_parameters machinery that could hang;
  * payload = 1 058 196 fp32 = 4.23 MB: a single latency-bound collective (ring over 7 xGMI links moves
    2*(7/8)*4.23 MB per GPU, ~50 us), so one bucket, issued right after backward on the compute stream's
    successor; nothing to gain from splitting it.
"""
import hashlib

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    def __init__(self, module, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.names = None
        self.params = None
        self.flat = None
        self.views = None
        self.always_pack = False      # tests: exercise the packing path in a single process

    def _build(self):
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad and p.grad is not None]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.world > 1:
            # every rank must have built the same bucket: compare a digest of (name, numel) lists
            digest = hashlib.sha1(repr([(n, p.numel()) for n, p in named]).encode()).digest()[:8]
            v = torch.tensor([int.from_bytes(digest, "little") % (2 ** 40), total], dtype=torch.float64, device=dev)
            lo, hi = v.clone(), v.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError("gradient buckets differ across ranks (different live-parameter sets)")

    @property
    def payload_bytes(self):
        return 0 if self.flat is None else self.flat.numel() * 4

    def reduce(self):
        """Average the gradients over the ranks in place.  Call after backward(), before optimizer.step().
        Single process: nothing to exchange, the gradients stay where autograd put them.  Otherwise the live gradients
        are packed into the flat bucket with one multi-tensor copy, all-reduced once, and the parameters' .grad are
        re-pointed at the bucket's slices (no copy back)."""
        if self.world == 1 and not self.always_pack:
            return
        if self.flat is None:
            self._build()
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        for i in missing:      # parameter lost its gradient this step (e.g. pre-training without the flow term): zeros
            self.views[i].zero_()
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        torch._foreach_copy_([self.views[i] for i in live], [self.params[i].grad for i in live])
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v


def broadcast_parameters(module, src=0, process_group=None):
    """Make every rank start from rank `src`'s parameters and buffers (incl. BN running statistics)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


def shard_batch(batch, rank, world):
    """Contiguous split of the leading (sample) dimension of every tensor: rank r gets samples [r*B/W, (r+1)*B/W)."""
    out = {}
    for k, v in batch.items():
        b = v.shape[0]
        assert b % world == 0, "batch %d not divisible by world size %d" % (b, world)
        out[k] = v[rank * (b // world):(rank + 1) * (b // world)]
    return out
