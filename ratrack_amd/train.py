"""Training step of the backbone (counterpart of the inner loop of main_utils.epoch, main_utils.py:122-156,
248-251, minus the host-side label/GT machinery): forward in train mode (batch-statistic BatchNorm; the training
path of ratrack_amd/train_path.py: fused HIP operators with hand-written backward kernels under autograd), multi-task
loss, backward, gradient all-reduce across ranks, optimizer step -- optionally the whole step as one hipGraph."""
import torch

from . import loss as L
from .ddp import FlatGradAllReducer


def make_optimizer(model, lr=1e-3, capturable=False):
    """Adam(lr=1e-3, weight_decay=1e-10) + StepLR(step=1, gamma=0.97) as main.py:61-62."""
    if capturable:      # a device-resident learning rate: the scheduler updates it in place, the captured graph reads it
        lr = torch.tensor(float(lr), device=next(model.parameters()).device)
    # fused: ONE multi-tensor kernel per step; the foreach implementation in capturable mode issues ~200 tiny kernels for the
    # per-parameter step counters and bias corrections
    on_gpu = next(model.parameters()).is_cuda
    if on_gpu:      # one launch for all parameters (ratrack_amd/optim.py): torch's fused Adam takes five
        from .optim import FusedAdam
        opt = FusedAdam(model.parameters(), lr=lr, weight_decay=1e-10)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-10, capturable=capturable, fused=on_gpu)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.97)
    return opt, sched


class Trainer:
    """graph=True: after `graph_warmup` eager steps the step is captured into hipGraphs and replayed; the batch is copied
    into static input buffers.  The step issues hundreds of kernels whose CPU launch cost rivals their GPU time at B=64; a
    replay is a single launch.
      * one process (world 1): ONE graph = forward, loss, backward, Adam;
      * data parallel (world > 1): graph A = forward, loss, backward, pack of the gradients into the flat bucket; the RCCL
        all-reduce of the bucket issued eagerly on the same stream (stream ordered, no host synchronisation); graph B = Adam
        on gradients that alias the bucket.  `graph_collective=True` captures the all-reduce as well (one graph) -- RCCL
        supports stream capture, but this build has only been run on one GPU, so it is opt-in.
    Shapes must stay fixed (the reference trains on fixed-size clouds, configs.yaml num_points) -- a different shape or
    loss configuration re-captures.  The returned loss items and GRU state are the graph's static output tensors: every
    replay overwrites them in place, so copy (`.clone()` / `float()`) what must outlive the next step."""

    def __init__(self, model, lr=1e-3, process_group=None, graph=False, graph_warmup=3, graph_collective=False, split_graph=None,
                 deterministic=False):
        """deterministic: every step is reproducible bit for bit -- gradients, losses, outputs, hence the whole trajectory from the same
        weights and batches (train_ops.set_deterministic: order-independent batch statistics and first-layer weight gradients; about
        3 % slower at B = 64; across ranks the RCCL all-reduce sums in its own fixed order).
        Range of the exact statistics (csrc/rtk_common.h rtk_stat_add): a forward addend must stay below 2^53 units of 2^-36, a backward
        addend (an element of dy or dy.x_hat) below 2^23 = 8.4e6 in magnitude; beyond that -- a gradient spike, a loss scale of 1e7 --
        the statistic reads NaN in this mode (the default mode's float64 atomics take such a step).  Addends are truncated towards zero
        at their unit (2^-36 / 2^-66), a downward bias far below fp32 rounding."""
        self.model = model
        self.deterministic = bool(deterministic)
        self._dev = next(model.parameters()).device
        if self._dev.type == "cuda":
            from . import train_ops
            train_ops.enable_zero_arena(self._dev)      # the step's ~200 small zero-initialised buffers: one fill
        self._one = torch.ones((), dtype=torch.float32, device=self._dev)
        self.opt, self.sched = make_optimizer(model, lr, capturable=graph)
        self.reducer = FlatGradAllReducer(model, process_group)
        self.graph = graph
        # split_graph: None = automatic (split when there is a collective that is not to be captured)
        self.split = (self.reducer.world > 1 and not graph_collective) if split_graph is None else bool(split_graph)
        self._warm = graph_warmup
        self._g = None
        self._g_opt = None
        self._static = None
        self._key = None

    def _forward_backward(self, *args):
        if self._dev.type != "cuda":
            return self._forward_backward_impl(*args)
        from . import train_ops
        prev = train_ops.set_deterministic(self.deterministic)
        try:
            return self._forward_backward_impl(*args)
        finally:
            train_ops.set_deterministic(prev)

    def _forward_backward_impl(self, pc1, pc2, feature1, feature2, gt_warp, gt_cls, h, n_valid, pretrain):
        if self._dev.type == "cuda":
            from . import train_ops
            self.opt.zero_grad(set_to_none=True)        # last step's gradients may be views of the arena that is cleared now
            train_ops.arena_begin_step(self._dev)
        if n_valid is not None:
            flow, h_out, cls, *_ = self.model.backbone(pc1, pc2, feature1, feature2, h, n_valid=n_valid)
        else:
            flow, h_out, cls, *_ = self.model.backbone(pc1, pc2, feature1, feature2, h)
        if self._dev.type == "cuda":
            total, items = train_ops.backbone_loss(pc1, flow, cls, gt_warp, gt_cls, pretrain=pretrain,      # one kernel, values + gradients
                                                   n_valid=None if n_valid is None else n_valid[0].contiguous())
        else:
            assert n_valid is None, "padded batches train on the GPU path"
            total, items = L.backbone_loss(pc1 + flow, cls, gt_warp, gt_cls, pretrain=pretrain)
        self.opt.zero_grad(set_to_none=True)
        if self._dev.type == "cuda":
            train_ops.begin_deferred_wgrads()           # the per-point layers queue their weight gradients ...
            try:
                total.backward(gradient=self._one)      # (a persistent seed: autograd would fill a new ones tensor every step)
            except BaseException:
                train_ops.drop_deferred_wgrads()        # a failed backward: nothing is launched on its half-built queue (the
                raise                                   # original error is the one the caller sees)
            train_ops.flush_deferred_wgrads()           # ... and they are issued together, eight per launch, and delivered to .grad
            train_ops.arena_end_step(self._dev)
        else:
            total.backward()
        self.reducer.pack()
        # detached: a caller holding last step's loss must not keep its autograd graph (and the parameters' gradient
        # accumulators, bound to the stream of that step) alive into the next step / into the graph capture
        return {k: v.detach() for k, v in items.items()}, h_out.detach()

    def _optimize(self):
        self.reducer.unpack()
        self.opt.step()
        self._invalidate_folded_engine()

    def _invalidate_folded_engine(self):
        """The parameters have just changed in place.  FusedAdam.step bumps their version counters when it runs eagerly; a graph
        replay runs no Python at all -- neither that bump nor anything in _optimize -- so step() calls this after every replay."""
        inv = getattr(self.model, "invalidate_fused", None)
        if inv is not None:
            inv()

    def _step(self, *args):
        out = self._forward_backward(*args)
        self.reducer.all_reduce()
        self._optimize()
        return out

    @staticmethod
    def _capture(fn):
        g = torch.cuda.CUDAGraph()
        import torch.distributed as dist
        # with a process group alive, its watchdog thread polls events while we capture: only this thread's calls may
        # invalidate the capture
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            out = fn()
        return g, out

    def step(self, pc1, pc2, feature1, feature2, gt_warp, gt_cls, h=None, pretrain=False, n_valid=None):
        """One optimisation step on this rank's shard.  Returns the loss items (python floats are NOT taken here:
        no device->host sync inside the step).
        n_valid (2,B) int32 on the device: a padded batch of clouds of different sizes (vod_gt.pad_frame_pairs; gt_warp / gt_cls
        padded alike) -- the counts live on the device, so ONE captured graph serves every batch of that padded shape: real
        radar frames (242 ... 352 points here, every consecutive pair of different sizes) train at B = 1 without re-capturing."""
        self.model.train()
        if n_valid is not None:
            assert n_valid.is_cuda and n_valid.dtype == torch.int32 and tuple(n_valid.shape) == (2, pc1.shape[0])
        args = [pc1, pc2, feature1, feature2, gt_warp, gt_cls, h, n_valid]
        if not self.graph:
            return self._step(*args, pretrain)
        key = tuple((tuple(t.shape), t.dtype) if t is not None else None for t in args) + (bool(pretrain),)
        if key != self._key:
            self._key, self._g, self._g_opt, self._count = key, None, None, 0
        if self._g is None and self._count < self._warm:      # eager warm-up (MIOpen finds its kernels, the bucket is built)
            self._count += 1
            return self._step(*args, pretrain)
        if self._g is None:
            self._static = [t.clone() if t is not None else None for t in args]
            torch.cuda.synchronize()
            self.opt.zero_grad(set_to_none=True)
            if self.split:
                self._g, self._out = self._capture(lambda: self._forward_backward(*self._static, pretrain))
                self._g_opt, _ = self._capture(self._optimize)
            else:
                self._g, self._out = self._capture(lambda: self._step(*self._static, pretrain))
        pairs = [(dst, src) for dst, src in zip(self._static, args) if dst is not None]
        same = [p for p in pairs if p[0].dtype == p[1].dtype]
        if same:                               # ONE multi-tensor copy into the static buffers instead of seven device-to-device memcpys
            torch._foreach_copy_([p[0] for p in same], [p[1] for p in same])
        for dst, src in pairs:
            if dst.dtype != src.dtype:
                dst.copy_(src)
        self._g.replay()
        if self._g_opt is not None:
            self.reducer.all_reduce()
            self._g_opt.replay()
        self._invalidate_folded_engine()
        return self._out
