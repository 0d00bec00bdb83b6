"""Data parallelism for the RaTrack backbone: one process per GPU, gradients averaged with ONE flat
all-reduce over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference's multi-GPU story is nn.DataParallel (models/model.py:38-41): single process, replicas
rebuilt every step, outputs gathered on GPU 0 -- unusable at its batch size of 1.  Frame-pairs are
independent in the backbone and BatchNorm statistics stay per replica (as under DataParallel), so the
only exchange of a training step is the gradient all-reduce:

  * 42 % of the parameters never receive a gradient on the backbone path.  The bucket
    holds the parameters that have EVER had a gradient after a backward (it grows, with a cross-rank digest
    check, when the loss configuration changes); the set is a property of the graph, identical on every
    rank -- there is no find_unused_parameters machinery that could hang;
  * payload = 1 058 196 fp32 = 4.23 MB: a single latency-bound collective (ring over 7 xGMI links moves
    2*(7/8)*4.23 MB per GPU, ~50 us), so one bucket, issued right after backward on the compute stream's
    successor; nothing to gain from splitting it.
"""
import hashlib

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """pack() -> all_reduce() -> unpack() (reduce() = the three in a row).  The split lets a trainer capture everything up to
    pack() in one hipGraph and the optimizer step after unpack() in a second one, with the RCCL call issued eagerly between
    the two replays (stream ordered, no host synchronisation)."""

    def __init__(self, module, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._candidates = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self._ids = set()
        self.names = None
        self.params = None
        self.flat = None
        self.views = None
        self.always_pack = False      # tests: exercise the packing path in a single process
        self.always_reduce = False    # tests: issue the collective even in a world of one (a 1-rank RCCL group on one GPU)

    def _build(self):
        """Bucket = every parameter that has EVER received a gradient on this rank (in named_parameters order).  The set is
        a property of the graph, identical on every rank, and verified across ranks (digest all-reduce) each time it grows
        -- e.g. when training switches from the segmentation-only pre-training loss (main_utils.py:148) to the full loss
        and the flow decoder's parameters start receiving gradients."""
        named = [(n, p) for n, p in self._candidates if p.grad is not None or id(p) in self._ids]
        self._ids = {id(p) for _, p in named}
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

    @property
    def active(self):
        return self.world > 1 or self.always_pack

    def pack(self):
        """Copy the live gradients into the flat bucket (one multi-tensor copy); parameters of the bucket without a
        gradient this step contribute zeros.  Grows the bucket when a parameter outside it shows up with a gradient."""
        if not self.active:
            return
        if self.flat is None or any(p.grad is not None and id(p) not in self._ids for _, p in self._candidates):
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the gradient bucket changed during stream capture: run an eager step with this loss "
                                   "configuration first")
            self._build()
        self._live = [i for i, p in enumerate(self.params) if p.grad is not None]
        for i, p in enumerate(self.params):
            if p.grad is None:      # no gradient this step (e.g. the flow decoder under the pre-training loss): zeros
                self.views[i].zero_()
        if self._live:
            torch._foreach_copy_([self.views[i] for i in self._live], [self.params[i].grad for i in self._live])

    def all_reduce(self):
        """ONE collective: sum over the ranks, then the mean."""
        if self.world > 1 or (self.always_reduce and self.flat is not None and dist.is_initialized()):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)

    def unpack(self):
        """Re-point the .grad of the parameters that had one at the bucket's slices (no copy back).  Parameters without a
        gradient keep None -- the optimizer skips them, as it does in the reference's single-process loop; the set is the
        same on every rank, so the replicas stay identical."""
        if not self.active:
            return
        for i in self._live:
            self.params[i].grad = self.views[i]

    def reduce(self):
        """Average the gradients over the ranks in place.  Call after backward(), before optimizer.step().
        Single process: nothing to exchange, the gradients stay where autograd put them."""
        self.pack()
        self.all_reduce()
        self.unpack()


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
