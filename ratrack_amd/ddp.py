"""Data parallelism for the RaTrack backbone: one process per GPU, gradients averaged with ONE flat
all-reduce over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference's multi-GPU story is nn.DataParallel (models/model.py:38-41): single process, replicas
rebuilt every step, outputs gathered on GPU 0 -- unusable at its batch size of 1.  Frame-pairs are
independent in the backbone and BatchNorm statistics stay per replica (as under DataParallel), so the
only exchange of a training step is the gradient all-reduce:

  * 42 % of the parameters never receive a gradient on the backbone path.  The bucket has a
    slot for every trainable parameter (1 816 031 fp32 = 7.26 MB, zeros where a parameter has no gradient), so its
    size never depends on a rank's loss configuration: no find_unused_parameters machinery, nothing that could hang;
    a digest of the live set rides in the same collective and is checked on the host once per configuration;
  * gradient payload = 1 058 196 fp32 = 4.23 MB of the 7.26 MB: a single latency-bound collective (a ring over 7 xGMI
    links moves 2*(7/8)*7.26 MB per GPU, ~85 us of bandwidth), so one bucket, issued right after backward on the
    compute stream's successor; nothing to gain from splitting it.
"""
import hashlib

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """pack() -> all_reduce() -> unpack() (reduce() = the three in a row).  The split lets a trainer capture everything up to
    pack() in one hipGraph and the optimizer step after unpack() in a second one, with the RCCL call issued eagerly between
    the two replays (stream ordered, no host synchronisation).

    The bucket has a slot for EVERY trainable parameter, in named_parameters order: its size is a property of the model, so
    every rank always enters the same collective with the same count -- whatever its loss configuration or data did (round 2
    grew the bucket with the set of live parameters and checked a digest in collectives of its own, which a single diverging
    rank entered alone: a hang instead of an error).  Parameters without a gradient this step contribute zeros and keep
    `grad = None` (Adam skips them, as in the reference's single-process loop).  GUARD trailing words carry a digest of this
    rank's live set; they are summed by the same all-reduce and compared with world x the local values afterwards, so replicas
    whose live sets differ raise before the optimizer step desynchronises them."""
    GUARD = 5      # four 16-bit pieces of the live-set digest (exact in fp32 sums over <= 256 ranks) + the live count

    def __init__(self, module, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.always_pack = False      # tests: exercise the packing path in a single process
        self.always_reduce = False    # tests: issue the collective even in a world of one (a 1-rank RCCL group on one GPU)
        self._buf = self.flat = self.views = None
        self._live, self._live_key, self._guard_dev, self._guard_host = [], None, None, None
        self._checked_key = None

    def _build(self):
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self._buf = torch.zeros(total + self.GUARD, dtype=torch.float32, device=dev)
        self.flat = self._buf[:total]
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def payload_bytes(self):
        """Bytes of gradient data in the collective (the live parameters'; the rest of the bucket are zeros)."""
        return 4 * sum(self.params[i].numel() for i in self._live)

    @property
    def bucket_bytes(self):
        return 0 if self._buf is None else self._buf.numel() * 4

    @property
    def active(self):
        return self.world > 1 or self.always_pack

    def pack(self):
        """Copy the live gradients into the bucket (one fill + one multi-tensor copy) and stamp the guard words."""
        if not self.active:
            return
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if self._buf is None:
            if capturing:
                raise RuntimeError("the gradient bucket does not exist yet: run an eager step before capturing")
            self._build()
        self._live = [i for i, p in enumerate(self.params) if p.grad is not None]
        key = tuple(self._live)
        if key != self._live_key:
            if capturing:
                raise RuntimeError("the set of parameters with a gradient changed during stream capture: run an eager step with this "
                                   "loss configuration first")
            digest = int.from_bytes(hashlib.sha1(repr([(self.names[i], self.params[i].numel()) for i in self._live]).encode()).digest()[:8], "little")
            g = torch.tensor([float((digest >> s) & 0xFFFF) for s in (0, 16, 32, 48)] + [float(len(self._live))], dtype=torch.float32)
            self._guard_host, self._guard_dev, self._live_key = g, g.to(self._buf.device), key
        self._buf.zero_()
        if self._live:
            torch._foreach_copy_([self.views[i] for i in self._live], [self.params[i].grad for i in self._live])
        self._buf[self.flat.numel():].copy_(self._guard_dev)

    def all_reduce(self):
        """ONE collective: sum over the ranks, then the mean."""
        if self.world > 1 or (self.always_reduce and self._buf is not None and dist.is_initialized()):
            world = dist.get_world_size(self.group)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)      # gradients + guard words
            capturing = self._buf.is_cuda and torch.cuda.is_current_stream_capturing()
            if not capturing and self._checked_key != self._live_key:
                # first eager all-reduce with this live set: every rank must have stamped the same guard (one small device->host
                # read, once per loss configuration -- the warm-up steps in front of a capture pay it)
                got, want = self._buf[self.flat.numel():].cpu(), self._guard_host * world
                if not torch.equal(got, want):
                    raise RuntimeError("gradient buckets differ across ranks (different live-parameter sets): guard %s, expected %s"
                                       % (got.tolist(), want.tolist()))
                self._checked_key = self._live_key
            self.flat.div_(world)

    def unpack(self):
        """Re-point the .grad of the parameters that had one at the bucket's slices (no copy back).  Parameters without a
        gradient keep None -- the optimizer skips them, as it does in the reference's single-process loop."""
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
