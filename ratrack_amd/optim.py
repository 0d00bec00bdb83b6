"""Adam for the train step in ONE kernel launch (csrc/train_optim.hip, rtk_adam_multi).

Same update rule, hyper-parameters and state layout (per-parameter 'step', 'exp_avg', 'exp_avg_sq') as torch.optim.Adam, which
the reference uses (main.py:61: Adam(lr, weight_decay=1e-10)); parameters without a gradient are skipped, as there.  torch's own
fused implementation walks the 130 parameter tensors of this model in five launches (0.11 ms per step: 3.5 % of the B = 1 step);
here a device table of {param, grad, exp_avg, exp_avg_sq, numel, first workgroup, step} rows drives a single grid.

The table is rebuilt whenever the set of (parameter, gradient) addresses changes.  Under stream capture it is filled by an
asynchronous copy from a pinned host buffer -- a memcpy node of the graph, replayed with the same contents."""
import ctypes

import torch

from . import _lib

_p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_lib.SIGNATURES.update({"rtk_adam_multi": [_i, _p, ctypes.c_long, _p, _f, _f, _f, _f, _f, _p, _p]})
CHUNK = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        assert len(self.param_groups) == 1, "FusedAdam: one parameter group (the reference has one)"
        self._key, self._blocks = None, 0
        dev = self.param_groups[0]["params"][0].device
        assert dev.type == "cuda", "FusedAdam drives a HIP kernel; use torch.optim.Adam on the CPU"
        self._steps = torch.zeros(len(self.param_groups[0]["params"]), dtype=torch.float32, device=dev)      # state['step'] of every parameter
        self._index = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
        self._ticket = torch.zeros((), dtype=torch.int32, device=dev)
        # table buffers for every parameter, allocated here: pinning or allocating during a stream capture would invalidate it
        n = len(self.param_groups[0]["params"])
        self._pinned = torch.empty((n, 7), dtype=torch.int64).pin_memory()
        self._table = torch.empty((n, 7), dtype=torch.int64, device=dev)

    def add_param_group(self, param_group):
        if getattr(self, "_steps", None) is not None:
            raise NotImplementedError("FusedAdam: one parameter group, fixed at construction (the device table is sized for it)")
        return super().add_param_group(param_group)

    def _state_of(self, p):
        st = self.state[p]
        slot = self._steps[self._index[id(p)]]
        if not st:
            st["step"] = slot
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not torch.is_tensor(st["step"]) or st["step"].data_ptr() != slot.data_ptr():
            # state that came in through load_state_dict (a torch.optim.Adam checkpoint, map_location='cpu', an int64 / float64
            # counter ...): the kernel reads the step through a device pointer -- copy the value into this optimizer's own slot
            slot.copy_(torch.as_tensor(st["step"], dtype=torch.float32).reshape(()))
            st["step"] = slot
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
        return st

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        assert len(self.param_groups) == 1, "one parameter group (the reference has one)"
        g = self.param_groups[0]
        live = [p for p in g["params"] if p.grad is not None]
        if not live:
            return None
        rows, blocks = [], 0
        for p in live:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and \
                p.grad.numel() == p.numel(), "FusedAdam: contiguous fp32 parameters and gradients"
            st = self._state_of(p)
            rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blocks,
                         st["step"].data_ptr()))
            blocks += (p.numel() + CHUNK - 1) // CHUNK
        key = tuple(rows)
        if key != self._key:
            # host rows -> pinned buffer -> device table, asynchronously on the current stream: under capture this is a memcpy node
            # of the graph, and replays re-read the same pinned rows.  (The pinned buffer is only rewritten when the addresses
            # change, i.e. never while a graph that reads it is being replayed with the same tensors.)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()      # an earlier asynchronous copy may still be reading the pinned rows
            self._pinned[:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
            self._table[:len(rows)].copy_(self._pinned[:len(rows)], non_blocking=True)
            self._key, self._blocks = key, blocks
        lr = g["lr"]
        lr_ptr, lr_val = (lr.data_ptr(), 0.0) if isinstance(lr, torch.Tensor) else (None, float(lr))
        _lib.call("rtk_adam_multi", len(rows), self._table.data_ptr(), self._blocks, lr_ptr, lr_val, float(g["betas"][0]),
                  float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._ticket.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        # the kernel wrote the parameters through raw pointers: bump their version counters, as an in-place torch op would have
        # (Track4D's folded eval engine watches them; a graph replay re-runs the kernel, not this line -- Trainer drops the engine)
        torch.autograd.graph.increment_version(live)
        return None
