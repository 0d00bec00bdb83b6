"""Autograd bindings of the training-mode kernels (include/rtk_train.h).

`bn_relu(z, bn, ...)` = nn.BatchNorm2d in training mode followed by ReLU (and, with pool=True, by the max over the
neighbourhood axis) -- the tail of every SharedMLP layer of the reference (lib/pytorch_utils.py:20-32,
lib/pointnet2_modules.py:44-47) -- on a de-duplicated, row-weighted tensor.  No CPU / eager fallback.
"""
import ctypes
import os

import torch

from . import _lib, fused

_i, _f, _d, _p = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
_LayerP = ctypes.POINTER(fused._Layer)
_lib.SIGNATURES.update({
    "rtk_cost_volume_train": [_i] * 3 + [_p] * 6 + [_LayerP, _LayerP, _p, _i, _p, _p, _p, _p, _p, _p],
    "rtk_cost_volume_bwd": [_i] * 3 + [_p] * 3 + [_LayerP, _LayerP, _p, _p, _i] + [_p] * 12 + [_p],
    "rtk_cost_volume_split_train": [_i] * 3 + [_p] * 10 + [_LayerP, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "rtk_cost_volume_bwd_split": [_i] * 3 + [_p] * 5 + [_LayerP, _p, _i] + [_p] * 13 + [_p],
    "rtk_pack_split_layer": [_i, _i, _p, _i, _p, _p, _p],
    "rtk_scatter_add_rows": [_i] * 4 + [_p] * 3 + [_p],
    "rtk_sa_first_layer": [_i] * 6 + [_p] * 4 + [_i] + [_p] * 3 + [_p],
    "rtk_group_inverse_index": [_i] * 3 + [_p] * 3 + [_p],
    "rtk_three_interpolate_grad_gather": [_i] * 4 + [_p] * 6 + [_p],
    "rtk_sa_first_layer_bwd": [_i] * 5 + [_p] * 6 + [_i, _p, _p],
    "rtk_conv_bn_fwd": [_i] * 6 + [_p] * 7 + [_p],
    "rtk_conv_wgrad": [_i] * 6 + [_p] * 5 + [ctypes.c_long, _p],
    "rtk_conv_bn_bwd": [_i] * 6 + [_p] * 6 + [_d, _i, _p, _p, _p],
    "rtk_train_group_geometry": [_i] * 5 + [_p] * 3 + [_i] + [_p] * 3 + [_p],
    "rtk_train_interp_weights": [_i] * 3 + [_p] * 5 + [_p],
    "rtk_train_row_weights": [_i] * 3 + [_p] * 2 + [_p],
    "rtk_gru_step_bwd": [_i] * 3 + [_p] * 15 + [_p],
    "rtk_gru_pack_params": [_i, _i, ctypes.POINTER(ctypes.c_void_p)] + [_p] * 6 + [_p],
    "rtk_gru_wgrad": [_i] * 3 + [_p] * 9 + [_p],
    "rtk_patch_cost_bwd": [_i, _i, _p, _p, _p, _i, _LayerP, _p, _p, _i, _p, _p, _p, _p, _p, _p],
    "rtk_patch_dfeat_gather": [_i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "rtk_bn_train_stats": [_i] * 5 + [_p] * 3 + [_p],
    "rtk_bn_relu_fwd": [_i] * 5 + [_p, _p, _i, _p, _p],
    "rtk_bn_relu_fwd_fin": [_i] * 5 + [_p, _p, _p, _i, _p, _p],
    "rtk_conv_bn_fwd_fin": [_i] * 6 + [_p] * 8 + [_p],
    "rtk_bn_relu_bwd_stats": [_i] * 5 + [_p, _p, _p, _i, _p, _p],
    "rtk_bn_relu_bwd_apply": [_i] * 5 + [_p] * 5 + [_d, _p, _i, _p, _p, _p],
    "rtk_bn_relu_bwd_small": [_i] * 4 + [_p] * 4 + [_d, _p, _p, _p, _i, _p],
    "rtk_train_point_weights": [_i] * 3 + [_p] * 3 + [_p],
})


class _IIJob(ctypes.Structure):          # rtk_inverse_index_job_t (include/rtk_train.h)
    _fields_ = [("n_src", ctypes.c_int), ("positions", ctypes.c_int), ("idx", ctypes.c_void_p), ("off", ctypes.c_void_p), ("inv", ctypes.c_void_p),
                ("live", ctypes.c_void_p), ("live_mult", ctypes.c_int)]


_lib.SIGNATURES.update({"rtk_group_inverse_index_multi": [_i, _i, ctypes.POINTER(_IIJob), _p]})


def group_inverse_index_multi(samples, jobs):
    """jobs: list of (n_src, positions, idx int32 (samples, positions), off int32 (samples, n_src + 1), inv int16 (samples, positions)
    [, live int32 (samples) or None, live_mult]): all inverse tables in one launch.  live: only the first live[s] * live_mult positions
    of sample s enter its table (rows nothing downstream reads: zero gradient)."""
    for lo in range(0, len(jobs), 12):
        part = jobs[lo:lo + 12]
        arr = (_IIJob * len(part))()
        for k, job in enumerate(part):
            n_src, P, idx, off, inv = job[:5]
            live, mult = (job[5], job[6]) if len(job) > 5 else (None, 1)
            arr[k].n_src, arr[k].positions, arr[k].idx, arr[k].off, arr[k].inv = n_src, P, idx.data_ptr(), off.data_ptr(), inv.data_ptr()
            arr[k].live, arr[k].live_mult = _ptr(live), mult
        _lib.call("rtk_group_inverse_index_multi", samples, len(part), arr, _stream())


class _PoolSrc(ctypes.Structure):        # rtk_pool_src_t (include/rtk_train.h)
    _fields_ = [("dout", ctypes.c_void_p), ("karg", ctypes.c_void_p), ("par", ctypes.c_void_p), ("sums2", ctypes.c_void_p),
                ("dgamma_dbeta", ctypes.c_void_p)]


_PoolP = ctypes.POINTER(_PoolSrc)
_lib.SIGNATURES.update({
    "rtk_bn_relu_pool_fwd_fin_arg": [_i] * 5 + [_p, _p, _p, _p, _p, _p, _p],
    "rtk_pool_bwd_stats_arg": [_i] * 4 + [_p] * 5 + [_p],
    "rtk_conv_wgrad_stats": [_i] * 6 + [_p, _PoolP, _p, _p, _p, _d, _p, _p, _p, _p, _p, _p, ctypes.c_long, _p],
    "rtk_conv_bn_bwd_apply": [_i] * 6 + [_p, _PoolP, _p, _p, _p, _p, _p, _d, _p, _p, _p],
})


class _PwOperand(ctypes.Structure):      # rtk_pw_operand_t (include/rtk_train.h)
    _fields_ = [("ptr", ctypes.c_void_p), ("sample_stride", ctypes.c_long), ("pitch", ctypes.c_int), ("channels", ctypes.c_int),
                ("layout", ctypes.c_int), ("col0", ctypes.c_int)]


_PwP = ctypes.POINTER(_PwOperand)


class _PwWgJob(ctypes.Structure):        # rtk_pw_wgrad_job_t (include/rtk_train.h)
    _fields_ = [("samples", ctypes.c_int), ("positions", ctypes.c_int), ("dz", _PwP), ("nsrc", ctypes.c_int), ("srcs", _PwP),
                ("dw", ctypes.c_void_p), ("w_pitch", ctypes.c_int), ("dbias", ctypes.c_void_p)]


_lib.SIGNATURES.update({
    "rtk_pw_wgrad_multi": [_i, ctypes.POINTER(_PwWgJob), _p, ctypes.c_long, _p],
    "rtk_pw_conv": [_i, _i, _i, _PwP, _i, _PwP, _p, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "rtk_pw_wgrad": [_i, _i, _PwP, _i, _PwP, _p, _i, _p, _p, ctypes.c_long, _p],
    "rtk_backbone_loss": [_i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p],
    "rtk_pack_weights": [_i, _p, _p],
    "rtk_weightnet_bwd": [ctypes.c_long, _i] + [_p] * 14 + [ctypes.c_long, _p],
})


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- zero arena ----------------------------------------------------------------------------------------------------------------
# The training operators need ~200 small zero-initialised buffers per step (float64 batch sums, atomically accumulated weight
# gradients).  As separate torch.zeros() calls that is ~200 fill kernels of a few microseconds each -- a tenth of the B = 1
# step.  With an arena (Trainer enables it) they are slices of ONE buffer that is cleared ONCE at the start of the step.
# Slices live until the next arena_begin_step(): long enough for the step's backward and optimizer (weight gradients handed to
# autograd may be arena views).  Without an arena (plain autograd use, tests) every request is a torch.zeros().
_ARENA = {}      # device -> [buffer, next free word, active, zeroed extent (words), demand of the current step (words)]


def enable_zero_arena(device, nbytes=64 << 20):
    dev = torch.device(device)
    if dev.type == "cuda" and dev not in _ARENA:
        _ARENA[dev] = [torch.zeros(nbytes // 8, dtype=torch.float64, device=dev), 0, False, 0, 0]


def arena_begin_step(device):
    """Clears as much of the arena as the previous step asked for (the first step of a shape therefore takes torch.zeros();
    a captured step bakes in the extent of the warm-up steps before it, which ran the same shapes)."""
    a = _ARENA.get(torch.device(device))
    if a is not None:
        a[3] = min(a[0].numel(), max(a[3], a[4]))
        if a[3]:
            a[0][:a[3]].zero_()
        a[1], a[2], a[4] = 0, True, 0


def arena_end_step(device):
    a = _ARENA.get(torch.device(device))
    if a is not None:
        a[2] = False


def _zeros(shape, dtype, device):
    a = _ARENA.get(device)
    n = 1
    for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= int(d)
    if a is not None and a[2]:
        words = (n * (8 if dtype == torch.float64 else 4) + 15) // 16 * 2          # float64 words, 16-byte granules
        a[4] += words
        if a[1] + words <= a[3]:                                                   # inside the part cleared for this step
            sl = a[0][a[1]:a[1] + words]
            a[1] += words
            return (sl if dtype == torch.float64 else sl.view(torch.float32))[:n].view(shape)
    return torch.zeros(shape, dtype=dtype, device=device)


def arena_zeros(n, dtype, device):
    """Zero-initialised 1-D workspace of a 4- or 8-byte dtype from the step arena (torch.zeros outside a training step)."""
    if dtype in (torch.float32, torch.float64):
        return _zeros((n,), dtype, device)
    assert dtype in (torch.int32,), dtype
    return _zeros((n,), torch.float32, device).view(dtype)


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _BNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, nbt, row_weight, count, groups, eps, momentum, pool):
        assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 4, "bn_relu: z must be a CUDA fp32 (S,C,rows,ns) tensor"
        z = z.contiguous()
        S_, C, rows, ns = z.shape
        if row_weight is not None:
            assert row_weight.shape == (S_, rows) and row_weight.dtype == torch.float32 and row_weight.is_contiguous()
        dev = z.device
        sums = _zeros((_stat_slots(), groups, C, 2), torch.float64, dev)
        _lib.call("rtk_bn_train_stats", S_, C, rows, ns, groups, z.data_ptr(), _ptr(row_weight), _sums_ptr(sums), _stream())
        par = torch.empty(4, groups, C, dtype=torch.float32, device=dev)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        fin = _BnFin(_sums_ptr(sums), float(count), g.data_ptr(), b.data_ptr(), float(eps), float(momentum), _ptr(running_mean),
                     _ptr(running_var), _ptr(nbt), None)
        y = torch.empty((S_, C, rows) if pool else (S_, C, rows, ns), dtype=torch.float32, device=dev)
        _lib.call("rtk_bn_relu_fwd_fin", S_, C, rows, ns, groups, z.data_ptr(), ctypes.byref(fin), par.data_ptr(), int(pool), y.data_ptr(),
                  _stream())      # finalisation + normalise + ReLU (+ max-pool) in one launch
        ctx.save_for_backward(z, par, row_weight)
        ctx.cfg = (count, groups, pool)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, par, row_weight = ctx.saved_tensors
        count, groups, pool = ctx.cfg
        S_, C, rows, ns = z.shape
        dy = dy.contiguous()
        dev = z.device
        sums2 = _zeros((_stat_slots(), groups, C, 2), torch.float64, dev)
        _lib.call("rtk_bn_relu_bwd_stats", S_, C, rows, ns, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), int(pool),
                  _sums_ptr(sums2), _stream())
        dz = torch.empty_like(z)
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        _lib.call("rtk_bn_relu_bwd_apply", S_, C, rows, ns, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), _ptr(row_weight),
                  _sums_ptr(sums2), float(count), None, int(pool), dz.data_ptr(), dgb.data_ptr(), _stream())
        return dz, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None


def bn_relu(z, bn, row_weight=None, count=None, groups=1, pool=False):
    """z (S,C,rows,ns) -> relu(batch_norm_train(z)) of shape z, or (S,C,rows) with pool=True (max over ns).
    bn: the nn.BatchNorm2d whose weight / bias / running statistics are used and updated exactly as its own
    training-mode forward would (momentum, unbiased running variance, num_batches_tracked).
    row_weight (S,rows) / count: statistics weights of de-duplicated rows and the reference element count per
    channel and group (default: all ones, S/groups * rows * ns).  groups > 1: separate statistics for consecutive
    batch slices, running statistics updated slice by slice (= the reference's sequential per-frame calls)."""
    S_, C, rows, ns = z.shape
    if count is None:
        count = (S_ // groups) * rows * ns
    momentum = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats
    return _BNReLU.apply(z, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                         bn.num_batches_tracked if track else None, row_weight, float(count), int(groups), bn.eps, momentum,
                         bool(pool))


# ---- per-point layers (virtual concatenation, both layouts) -------------------------------------------------------------------

def _pw_layout(t):
    """(S,C,P) tensor -> 0 (channel-major planes, positions contiguous) / 1 (point-major rows, channels contiguous) / None."""
    if t.stride(2) == 1 or t.shape[2] == 1:
        return 0
    if t.stride(1) == 1 or t.shape[1] == 1:
        return 1
    return None


def _pw_tensor(t):
    """A (S,C,P) fp32 CUDA tensor the kernels can address directly (either layout, any pitch); copies only exotic strides."""
    assert t.dim() == 3 and t.is_cuda and t.dtype == torch.float32, "per-point layers take (S,C,P) CUDA fp32 tensors"
    return t if _pw_layout(t) is not None else t.contiguous()


def _pw_operands(tensors, cols):
    arr = (_PwOperand * len(tensors))()
    for i, (t, c0) in enumerate(zip(tensors, cols)):
        lay = _pw_layout(t)
        pitch = (t.stride(1) if t.shape[1] > 1 else t.shape[2]) if lay == 0 else (t.stride(2) if t.shape[2] > 1 else t.shape[1])
        arr[i].ptr, arr[i].sample_stride, arr[i].pitch = t.data_ptr(), t.stride(0), pitch
        arr[i].channels, arr[i].layout, arr[i].col0 = t.shape[1], lay, c0
    return arr


def _pw_like(t, channels=None, point_major=None):
    """Uninitialised (S,C,P) tensor in t's layout (or the requested one)."""
    S_, C, P = t.shape
    C = channels if channels is not None else C
    pm = (_pw_layout(t) == 1) if point_major is None else point_major
    if pm:
        return torch.empty(S_, P, C, dtype=torch.float32, device=t.device).permute(0, 2, 1)
    return torch.empty(S_, C, P, dtype=torch.float32, device=t.device)


def _pw_forward(srcs, cols, W, bias, out, row_w=None, groups=1, sums=None):
    S_, _, P = srcs[0].shape
    _lib.call("rtk_pw_conv", S_, P, len(srcs), _pw_operands(srcs, cols), 1, _pw_operands([out], [0]), W.data_ptr(), W.stride(0), 0,
              _ptr(bias), 0, _ptr(row_w), groups, (_sums_ptr(sums) if sums is not None else None), out.shape[1], _stream())


STAT_SLOTS, STAT_SLOTS_ORDERED = 8, 16      # RTK_STAT_SLOTS / RTK_STAT_SLOTS_ORDERED (include/rtk_train.h): float64 words per batch statistic
DETERMINISTIC = False      # set_deterministic()


def set_deterministic(on=True):
    """Reproducible training: every sum of a step that used to depend on the order in which workgroups arrive is accumulated
    order-independently -- the batch statistics as exact fixed-point limbs (the tag in bit 0 of the buffers' pointers,
    csrc/rtk_common.h rtk_stat_add), the offset columns of the first layers' weight gradients as per-sample shares added in a fixed
    order (rtk_sa_first_layer_bwd's dwx_ws); everything else in the step is order-independent in either mode.  Gradients, losses and
    outputs are then bit-identical from run to run (tests/test_train_gpu.py, tools/hazard_train.py); the step takes about 3 % longer
    at B = 64.  Off (the default): float64 / float atomics there, gradients differ from run to run at 1e-6 of a tensor's largest
    element.  Returns the previous setting.  (A captured step keeps the mode it was captured in.)"""
    global DETERMINISTIC
    prev, DETERMINISTIC = DETERMINISTIC, bool(on)
    return prev


def _stat_slots():
    return STAT_SLOTS_ORDERED if DETERMINISTIC else STAT_SLOTS


def _sums_ptr(t):
    """Device pointer of a batch-statistics buffer with the accumulation mode in bit 0 (include/rtk_train.h RTK_STAT_SLOTS)."""
    p = t.data_ptr()
    assert p % 8 == 0
    return p | 1 if DETERMINISTIC else p
_WGRAD_WS = 4 << 20        # floats: 1024 partial 64 x 64 blocks (rtk_pw_wgrad splits the position axis as far as this allows)


# The weight gradients of a training step are leaves of its graph -- only the optimizer reads them.  Between begin_deferred_wgrads()
# and flush_deferred_wgrads() (train.Trainer brackets the backward with them) the per-point layers queue theirs instead of launching
# them: 35 launches of 10-20 us, each filling the chip for a few microseconds, become a handful of launches with eight jobs each
# (rtk_pw_wgrad_multi).  A queued gradient does not travel through autograd -- the engine may copy what a backward returns, and would
# copy it before it is computed -- the flush assigns / adds it to the parameter's .grad itself.
_DEFERRED = None


WGRAD_TWO_STREAMS = True
_WGRAD_SIDE = {}


def _split_wgrad_jobs(jobs):
    """Chunks of eight jobs (one launch each) dealt alternately to two streams; jobs that accumulate into the same dW -- a parameter
    used twice -- stay in their order on ONE stream (the sums keep their order).  None if a chunk would need both streams."""
    owner, out = {}, ([], [])
    for ci, k in enumerate(range(0, len(jobs), 8)):
        chunk = jobs[k:k + 8]
        prefs = {owner[j[3].data_ptr()] for j in chunk if j[3].data_ptr() in owner}
        if len(prefs) > 1:
            return None
        which = prefs.pop() if prefs else ci % 2
        for j in chunk:
            owner[j[3].data_ptr()] = which
        out[which].extend(chunk)
    return out if out[0] and out[1] else None


def begin_deferred_wgrads():
    global _DEFERRED
    _DEFERRED = {"jobs": [], "assign": []}


def _run_wgrad_jobs(jobs):
    """jobs: [(dz (S,Co,P), srcs, cols, dW, dbias or None)]: dW[:, cols_i + k] += sum dz x src_i, dbias += sum dz -- one call."""
    arr = (_PwWgJob * len(jobs))()
    keep = []
    for k, (dz, srcs, cols, dW, dbias) in enumerate(jobs):
        S_, _, P = dz.shape
        a, b = _pw_operands([dz], [0]), _pw_operands(srcs, cols)
        keep += [a, b]
        arr[k].samples, arr[k].positions, arr[k].dz, arr[k].nsrc, arr[k].srcs = S_, P, a, len(srcs), b
        arr[k].dw, arr[k].w_pitch, arr[k].dbias = dW.data_ptr(), dW.stride(0), _ptr(dbias)
    ws = torch.empty(_WGRAD_WS, dtype=torch.float32, device=jobs[0][0].device)          # workgroup partials (uninitialised scratch)
    _lib.call("rtk_pw_wgrad_multi", len(jobs), arr, ws.data_ptr(), _WGRAD_WS, _stream())


def drop_deferred_wgrads():
    """End the deferral WITHOUT launching (the backward that was filling the queue failed)."""
    global _DEFERRED
    _DEFERRED = None


def flush_deferred_wgrads():
    """Issue the queued weight gradients and hand them to their parameters.  Always ends the deferral.
    The queued gradients bypass autograd's accumulator, so tensor hooks / post-accumulate-grad hooks on those parameters would
    never fire: refused when the gradient is queued (inside the backward, whose failure drops the queue: no half-finished step)
    rather than silently skipped.  (Queued dz / source tensors stay alive until this call.)"""
    global _DEFERRED
    q, _DEFERRED = _DEFERRED, None
    if not q or not q["jobs"]:
        return
    jobs = q["jobs"]
    if jobs[0][0].is_cuda:
        # explicit ordering (round-5 advice): a job queued by a backward that ran on one of the MSG side streams (train_path._msg_scales)
        # has its dz / source tensors produced THERE; the flush reads them on the current stream (and the weight-gradient side stream,
        # which forks from it below).  The autograd engine's end-of-backward stream sync happens to order the two today; this does not
        # depend on it (stream waits are a few hundred nanoseconds each and graph-capturable).
        from . import train_path as _TP
        cur = torch.cuda.current_stream()
        capturing = torch.cuda.is_current_stream_capturing()
        for (dev_, _key), s_ in list(_TP._MSG_SIDE.items()):
            if dev_ != jobs[0][0].device:
                continue
            with torch.cuda.stream(s_):
                same = torch.cuda.is_current_stream_capturing() == capturing
            if same:      # (a side stream this capture never forked has no work of this step, and may not be joined from inside it)
                cur.wait_stream(s_)
    halves = _split_wgrad_jobs(jobs) if WGRAD_TWO_STREAMS and len(jobs) >= 16 and jobs[0][0].is_cuda else None
    if halves is not None:
        # the launches of the flush (eight jobs each) are independent of each other: every second one goes to a side stream
        # (train step 7.05 -> 6.97 ms at B = 64)
        cur = torch.cuda.current_stream()
        side = _WGRAD_SIDE.setdefault(jobs[0][0].device, torch.cuda.Stream(device=jobs[0][0].device))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _run_wgrad_jobs(halves[1])
        _run_wgrad_jobs(halves[0])
        cur.wait_stream(side)
    else:
        _run_wgrad_jobs(jobs)
    for param, grad in q["assign"]:
        g = grad.view_as(param)
        if param.grad is None:
            param.grad = g
        else:
            param.grad = param.grad + g      # (out of place: the existing gradient may be a view of a shared buffer)


def _pw_backward(ctx_needs, srcs, cols, W, dz, want_bias, dW=None, owner=None):
    """-> (dW (full shape of W, zero outside the used columns; accumulated into the given zero-initialised dW if any), dbias or
    None, [dsrc_i or None], deferred).  owner = (weight parameter, bias parameter or None): while weight gradients are being
    deferred, dW / dbias are only queued -- they hold their values after flush_deferred_wgrads(), which also delivers them to the
    owners' .grad -- and `deferred` tells the caller to return None for them to autograd."""
    S_, Co, P = dz.shape
    dz = _pw_tensor(dz)
    if dW is None:
        buf = _zeros((W.shape[0] * W.shape[1] + (Co if want_bias else 0),), torch.float32, dz.device)      # dW | dbias
        dW = buf[:W.shape[0] * W.shape[1]].view(W.shape[0], W.shape[1])
        dbias = buf[W.shape[0] * W.shape[1]:] if want_bias else None
    else:
        dbias = _zeros((Co,), torch.float32, dz.device) if want_bias else None
    deferred = _DEFERRED is not None and owner is not None and owner[0] is not None and (dbias is None or owner[1] is not None)
    if deferred:
        for param in owner:
            if param is not None and (param._backward_hooks or getattr(param, "_post_accumulate_grad_hooks", None)):
                raise RuntimeError("deferred weight gradients are delivered to .grad directly: gradient hooks on the per-point layers' "
                                   "parameters do not fire (run the backward outside begin_deferred_wgrads() to use hooks)")
        _DEFERRED["jobs"].append((dz, list(srcs), list(cols), dW, dbias))
        _DEFERRED["assign"].append((owner[0], dW))
        if dbias is not None:
            _DEFERRED["assign"].append((owner[1], dbias))
    else:
        _run_wgrad_jobs([(dz, srcs, cols, dW, dbias)])
    dsrcs = [None] * len(srcs)
    todo = [i for i in range(len(srcs)) if ctx_needs[i]]
    if todo:
        outs = [_pw_like(srcs[i]) for i in todo]
        _lib.call("rtk_pw_conv", S_, P, 1, _pw_operands([dz], [0]), len(outs), _pw_operands(outs, [cols[i] for i in todo]), W.data_ptr(),
                  W.stride(0), 1, None, 0, None, 1, None, 0, _stream())
        for i, o in zip(todo, outs):
            dsrcs[i] = o
    return dW, dbias, dsrcs, deferred


def _leaf(t):
    """The parameter a deferred gradient can be delivered to: t itself if it is a leaf that wants a gradient."""
    return t if (t is not None and t.is_leaf and t.requires_grad) else None


class _PwLinear(torch.autograd.Function):
    """z = W[:, cols] . [src_0 ; src_1 ; ...] (+ bias): a 1x1 convolution / nn.Linear over the channel axis of per-point tensors,
    the concatenation virtual (rtk_pw_conv); backward = rtk_pw_wgrad (weight + bias gradient, one launch) + rtk_pw_conv with the
    transposed weight (all input gradients, one launch)."""

    @staticmethod
    def forward(ctx, W, bias, cfg, *srcs):
        cols, out_pm = cfg
        srcs = [_pw_tensor(t) for t in srcs]
        W2 = W.detach().reshape(W.shape[0], -1)
        assert W2.stride(1) == 1
        out = _pw_like(srcs[0], W2.shape[0], point_major=out_pm)
        _pw_forward(srcs, cols, W2, bias.detach().contiguous() if bias is not None else None, out)
        ctx.save_for_backward(W, *srcs)
        ctx.cfg = (cols, bias is not None)
        ctx.owner = (_leaf(W), _leaf(bias))
        return out

    @staticmethod
    def backward(ctx, dz):
        W, *srcs = ctx.saved_tensors
        cols, has_bias = ctx.cfg
        W2 = W.detach().reshape(W.shape[0], -1)
        dW, dbias, dsrcs, deferred = _pw_backward(ctx.needs_input_grad[3:], srcs, cols, W2, dz, has_bias and ctx.needs_input_grad[1],
                                                  owner=ctx.owner if ctx.needs_input_grad[0] else None)
        if deferred:
            return (None, None, None) + tuple(dsrcs)
        return (dW.view_as(W) if ctx.needs_input_grad[0] else None, dbias, None) + tuple(dsrcs)


def _pw_cols(srcs, cols):
    if cols is not None:
        return list(cols)
    out, c = [], 0
    for t in srcs:
        out.append(c)
        c += t.shape[1]
    return out


def pw_linear(srcs, weight, bias=None, cols=None, out_point_major=False):
    """srcs: (S,Ci,P) tensors, channel-major or point-major views; weight (Co, K[,1,1]) with the sources' columns starting at `cols`
    (default: consecutive from 0); -> (S,Co,P) (a permuted view of an (S,P,Co) tensor with out_point_major)."""
    return _PwLinear.apply(weight, bias, (_pw_cols(srcs, cols), bool(out_point_major)), *srcs)


class _PwBnRelu(torch.autograd.Function):
    """relu(BatchNorm_train(W . [src_0 ; src_1 ; ...])) for per-point tensors: conv + batch sums in one kernel (rtk_pw_conv),
    finalize, normalise + ReLU (rtk_bn_relu_fwd); backward: the two BatchNorm passes, then as _PwLinear."""

    @staticmethod
    def forward(ctx, W, gamma, beta, cfg, *srcs):
        bn, row_w, count, groups, cols, gcounts = cfg
        srcs = [_pw_tensor(t) for t in srcs]
        W2 = W.detach().reshape(W.shape[0], -1)
        S_, _, P = srcs[0].shape
        Co = W2.shape[0]
        dev = srcs[0].device
        sums = _zeros((_stat_slots(), groups, Co, 2), torch.float64, dev)
        z = torch.empty(S_, Co, P, dtype=torch.float32, device=dev)
        _pw_forward(srcs, cols, W2, None, z, row_w, groups, sums)
        fin, par, _keep = _bn_fin(bn, sums, count, groups, gcounts)
        y = torch.empty_like(z)
        _lib.call("rtk_bn_relu_fwd_fin", S_, Co, P, 1, groups, z.data_ptr(), fin, par.data_ptr(), 0, y.data_ptr(), _stream())
        ctx.save_for_backward(W, z, par, row_w, gcounts, *srcs)
        ctx.owner = (_leaf(W), None)
        ctx.cfg = (count, groups, cols)
        return y

    @staticmethod
    def backward(ctx, dy):
        W, z, par, row_w, gcounts, *srcs = ctx.saved_tensors
        count, groups, cols = ctx.cfg
        S_, Co, P = z.shape
        dev = z.device
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        if S_ * P <= 65536 and P % 4 == 0:      # a channel fits one workgroup: statistics and apply in one launch
            split = groups == 2 and Co <= 128      # (two-frame batch: a workgroup per (channel, group), sums added to a zeroed dgb)
            dgb = _zeros((2, Co), torch.float32, dev) if split else torch.empty(2, Co, dtype=torch.float32, device=dev)
            _lib.call("rtk_bn_relu_bwd_small", S_, Co, P, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), _ptr(row_w), float(count),
                      _ptr(gcounts), dz.data_ptr(), dgb.data_ptr(), int(split), _stream())
        else:
            dgb = torch.empty(2, Co, dtype=torch.float32, device=dev)
            sums2 = _zeros((_stat_slots(), groups, Co, 2), torch.float64, dev)
            _lib.call("rtk_bn_relu_bwd_stats", S_, Co, P, 1, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), 0, _sums_ptr(sums2), _stream())
            _lib.call("rtk_bn_relu_bwd_apply", S_, Co, P, 1, groups, z.data_ptr(), dy.data_ptr(), par.data_ptr(), _ptr(row_w), _sums_ptr(sums2),
                      float(count), _ptr(gcounts), 0, dz.data_ptr(), dgb.data_ptr(), _stream())
        W2 = W.detach().reshape(W.shape[0], -1)
        dW, _, dsrcs, deferred = _pw_backward(ctx.needs_input_grad[4:], srcs, cols, W2, dz, False, owner=ctx.owner)
        return (None if deferred else dW.view_as(W), dgb[0], dgb[1], None) + tuple(dsrcs)


def pw_bn_relu(srcs, weight, bn, row_weight=None, count=None, groups=1, cols=None, group_counts=None):
    """relu(bn(conv1x1(cat(srcs)))) in training mode; bn: the nn.BatchNorm2d whose parameters / running statistics are used and
    updated; row_weight (S,P), count: as bn_relu; group_counts: optional device float64 (groups,) per-group element counts that
    replace `count` (padded batches of clouds of different sizes: rtk_train_point_weights).  -> (S,Co,P) channel-major."""
    S_, _, P = srcs[0].shape
    if count is None:
        count = (S_ // groups) * P
    return _PwBnRelu.apply(weight, bn.weight, bn.bias, (bn, row_weight, float(count), int(groups), _pw_cols(srcs, cols), group_counts), *srcs)


# ---- global feature appended to every point ---------------------------------------------------------------------------------

_lib.SIGNATURES.update({"rtk_gmax_cat_fwd": [_i, _i, _i, _p, _p, _p, _p], "rtk_gmax_cat_bwd": [_i, _i, _i, _p, _p, _p, _p]})


class _GmaxCat(torch.autograd.Function):
    """(S,C,N) -> (S,2C,N) = cat(f, max over the points broadcast) (models/track4d.py:92-95), one kernel each way."""

    @staticmethod
    def forward(ctx, f):
        f = f.contiguous()
        S_, C, N = f.shape
        out = torch.empty(S_, 2 * C, N, dtype=torch.float32, device=f.device)
        arg = torch.empty(S_, C, dtype=torch.int32, device=f.device)
        _lib.call("rtk_gmax_cat_fwd", S_, C, N, f.data_ptr(), out.data_ptr(), arg.data_ptr(), _stream())
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, = ctx.saved_tensors
        dout = dout.contiguous()
        S_, C2, N = dout.shape
        df = torch.empty(S_, C2 // 2, N, dtype=torch.float32, device=dout.device)
        _lib.call("rtk_gmax_cat_bwd", S_, C2 // 2, N, dout.data_ptr(), arg.data_ptr(), df.data_ptr(), _stream())
        return df


def gmax_cat(f):
    """f (S,C,N) fp32 on the GPU -> cat((f, f.max(-1)[0][:, :, None].expand(-1, -1, N)), dim=1)."""
    return _GmaxCat.apply(f)


# ---- SharedMLP chain of one set-abstraction scale ------------------------------------------------------------------------

class _BnFin(ctypes.Structure):          # rtk_bn_fin_t (include/rtk_train.h)
    _fields_ = [("sums", ctypes.c_void_p), ("count", ctypes.c_double), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p),
                ("num_batches_tracked", ctypes.c_void_p), ("group_counts", ctypes.c_void_p)]


def _bn_fin(bn, sums, count, groups, group_counts=None):
    """(rtk_bn_fin_t by reference, par buffer): the BatchNorm of `sums` is finalised by the kernel that consumes it (one launch less
    per layer and step than a separate finalisation kernel); par is written for the backward."""
    C = bn.num_features
    par = torch.empty(4, groups, C, dtype=torch.float32, device=sums.device)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats
    f = _BnFin(_sums_ptr(sums), float(count), bn.weight.detach().data_ptr(), bn.bias.detach().data_ptr(), float(bn.eps), float(momentum),
               _ptr(bn.running_mean if track else None), _ptr(bn.running_var if track else None),
               _ptr(bn.num_batches_tracked if track else None), _ptr(group_counts))
    return ctypes.byref(f), par, f


class _SumsPool:
    """The float64 statistics buffers of one chain pass, zeroed with ONE fill: call with a channel count to get the next
    (slots, groups, C, 2) slice."""

    def __init__(self, groups, channels, device):
        self.groups = groups
        self.slots = _stat_slots()
        self.buf = _zeros((self.slots * groups * 2 * sum(channels),), torch.float64, device)
        self.off = 0

    def __call__(self, c):
        n = self.slots * self.groups * c * 2
        t = self.buf[self.off:self.off + n].view(self.slots, self.groups, c, 2)
        self.off += n
        return t


class _SAChain(torch.autograd.Function):
    """[BN + ReLU -> 1x1 conv]* -> BN + ReLU -> max over the neighbourhood, starting from the first layer's
    pre-activation z1 (S,C1,rows,ns).  Each inner layer is ONE kernel forward (previous BatchNorm + ReLU on load, MFMA,
    this layer's batch sums in the epilogue) and TWO backward (statistics, apply; both recompute W^T dz) + a batched GEMM
    for the weight gradient; the normalised activations are stored once for that GEMM, nothing else is materialised."""

    @staticmethod
    def forward(ctx, w0, idx, dxyz, row_w, count, groups, bns, inv, nfeat, *tensors):
        # tensors = feats_0 .. feats_{nfeat-1}, g0, b0, W1, g1, b1, W2, g2, b2.  w0 (C1, 3 + C[,1,1]) = [offset columns | feature columns]
        feats = [_pw_tensor(t) for t in tensors[:nfeat]]
        tensors = tensors[nfeat:]
        S_, _, n_src = feats[0].shape
        W0 = w0.detach().reshape(w0.shape[0], -1)
        C1 = W0.shape[0]
        cols, c = [], 3
        for f in feats:
            cols.append(c)
            c += f.shape[1]
        dev = feats[0].device
        proj = torch.empty(S_, C1, n_src, dtype=torch.float32, device=dev)      # per-POINT projection by the feature columns
        _pw_forward(feats, cols, W0, None, proj)
        wx = W0[:, :3]
        _, rows, ns = idx.shape
        L = len(bns)
        weights = [None] + [tensors[3 * i - 1] for i in range(1, L)]
        f64 = _SumsPool(groups, [C1] + [w.shape[0] for w in weights[1:]], dev)
        sums = f64(C1)
        z1 = torch.empty(S_, C1, rows, ns, dtype=torch.float32, device=dev)
        _lib.call("rtk_sa_first_layer", S_, C1, rows, ns, groups, n_src, proj.data_ptr(), idx.data_ptr(), dxyz.data_ptr(), W0.data_ptr(),
                  W0.stride(0), _ptr(row_w), z1.data_ptr(), _sums_ptr(sums), _stream())      # the offset columns = the first three of W0
        # every BatchNorm is finalised by the kernel that consumes it (the next convolution, the pooling pass at the end)
        zs, ys, pars = [z1], [], []
        fin, par, _keep = _bn_fin(bns[0], sums, count, groups)
        for i in range(1, L):
            W = weights[i]
            Co, Ci = W.shape[0], W.shape[1]
            wc = W.detach().contiguous()
            z = torch.empty(S_, Co, rows, ns, dtype=torch.float32, device=dev)
            sums = f64(Co)
            _lib.call("rtk_conv_bn_fwd_fin", S_, Ci, Co, rows, ns, groups, zs[-1].data_ptr(), fin, par.data_ptr(), wc.data_ptr(), z.data_ptr(),
                      None, _ptr(row_w), _sums_ptr(sums), _stream())      # the normalised input is NOT stored (rtk_conv_wgrad recomputes it)
            pars.append(par)
            fin, par, _keep = _bn_fin(bns[i], sums, count, groups)
            zs.append(z)
        C = zs[-1].shape[1]
        out = torch.empty(S_, C, rows, dtype=torch.float32, device=dev)
        # the pooled layer also records which element of every row its gradient will go to (and that element's z): the backward
        # then needs no pass of its own over z for this layer
        zarg = torch.empty(S_, C, rows, dtype=torch.float32, device=dev)
        karg = torch.empty(S_, C, rows, dtype=torch.uint8, device=dev)
        _lib.call("rtk_bn_relu_pool_fwd_fin_arg", S_, C, rows, ns, groups, zs[-1].data_ptr(), fin, par.data_ptr(), out.data_ptr(), zarg.data_ptr(),
                  karg.data_ptr(), _stream())
        pars.append(par)
        ctx.save_for_backward(row_w, idx, dxyz, zarg, karg, *zs, *pars, *[w for w in weights[1:]], w0, *feats)
        ctx.bn_affine = [(b.weight.detach(), b.bias.detach()) for b in bns]
        ctx.cfg = (count, groups, L, n_src, nfeat, cols)
        ctx.inv = inv
        ctx.owner = (_leaf(w0), None)
        return out

    @staticmethod
    def backward(ctx, dout):
        count, groups, L, n_src, nfeat, cols = ctx.cfg
        saved = list(ctx.saved_tensors)
        feats, w0 = saved[len(saved) - nfeat:], saved[len(saved) - nfeat - 1]
        saved = saved[:len(saved) - nfeat - 1]
        row_w, idx, dxyz, zarg, karg = saved[0:5]
        saved = saved[4:]
        zs = saved[1:1 + L]
        pars, weights = saved[1 + L:1 + 2 * L], [None] + saved[1 + 2 * L:]
        W0 = w0.detach().reshape(w0.shape[0], -1)
        S_, _, rows, ns = zs[0].shape
        dev = dout.device
        dout = dout.contiguous()
        f64 = _SumsPool(groups, [z.shape[1] for z in zs], dev)
        # last layer (BatchNorm + ReLU + max-pool): its statistics from the recorded arg-max -- no pass over z --, its dz formed on load
        # by the two consumers below (`pool` source)
        C = zs[-1].shape[1]
        sums_last = f64(C)
        _lib.call("rtk_pool_bwd_stats_arg", S_, C, rows, groups, dout.data_ptr(), zarg.data_ptr(), karg.data_ptr(), pars[-1].data_ptr(),
                  _sums_ptr(sums_last), _stream())
        dgb_last = torch.empty(2, C, dtype=torch.float32, device=dev)
        pool = _PoolSrc(dout.data_ptr(), karg.data_ptr(), pars[-1].data_ptr(), _sums_ptr(sums_last), dgb_last.data_ptr())
        grads = {L - 1: (None, dgb_last[0], dgb_last[1])}
        dwbuf = _zeros((sum(w.numel() for w in weights[1:]) + W0.numel(),), torch.float32, dev)      # all dW of the chain | dW0
        dW0 = dwbuf[dwbuf.numel() - W0.numel():].view(W0.shape[0], W0.shape[1])
        dwoff = 0
        src, src_pool = zs[-1], ctypes.byref(pool)             # the gradient source of the layer being walked: (z, pool) or (dz, None)
        for i in range(L - 1, 0, -1):
            W = weights[i]
            Co, Ci = W.shape[0], W.shape[1]
            dW = dwbuf[dwoff:dwoff + W.numel()].view_as(W)
            dwoff += W.numel()
            wc = W.detach().contiguous()
            ga, be = ctx.bn_affine[i - 1]
            ws = torch.empty(2 * max(S_, 1024) * Ci * Co, dtype=torch.float32, device=dev)      # workgroup partials (uninitialised scratch)
            sums2 = f64(Ci)
            # ONE pass over (dz, z[i-1]): the weight gradient AND the statistics of the previous BatchNorm's backward
            _lib.call("rtk_conv_wgrad_stats", S_, Ci, Co, rows, ns, groups, src.data_ptr(), src_pool, zs[i - 1].data_ptr(), pars[i - 1].data_ptr(),
                      _ptr(row_w), float(count), wc.data_ptr(), ga.data_ptr(), be.data_ptr(), dW.data_ptr(), _sums_ptr(sums2), ws.data_ptr(),
                      ws.numel(), _stream())
            dzp = torch.empty_like(zs[i - 1])
            dgb = torch.empty(2, Ci, dtype=torch.float32, device=dev)
            _lib.call("rtk_conv_bn_bwd_apply", S_, Ci, Co, rows, ns, groups, src.data_ptr(), src_pool, wc.data_ptr(), zs[i - 1].data_ptr(),
                      pars[i - 1].data_ptr(), _ptr(row_w), _sums_ptr(sums2), float(count), dzp.data_ptr(), dgb.data_ptr(), _stream())
            grads[i] = (dW, grads[i][1], grads[i][2])
            grads[i - 1] = (None, dgb[0], dgb[1])
            src, src_pool = dzp, None
        dz = src
        flat = [grads[0][1], grads[0][2]]
        for i in range(1, L):
            flat += [grads[i][0], grads[i][1], grads[i][2]]
        # first layer: z1 = (Wf feats)[idx] + Wx.dxyz  ->  dproj = gather-sum of dz over each source point's positions, the offset
        # columns of dW0 = sum dz dxyz^T (same kernel), then the projection's own backward: feature columns of dW0, dfeats
        C1 = dz.shape[1]
        dproj = torch.empty(S_, C1, n_src, dtype=torch.float32, device=dev)
        if ctx.inv is not None:
            off, inv = ctx.inv[0], ctx.inv[1]
            if len(ctx.inv) > 2 and ctx.inv[2] is not None:        # table built on the geometry stream (TrainGeometry)
                torch.cuda.current_stream().wait_event(ctx.inv[2])
            dwx_ws = torch.empty(S_ * C1 * 3, dtype=torch.float32, device=dev) if DETERMINISTIC else None      # the samples' shares of the offset columns
            _lib.call("rtk_sa_first_layer_bwd", S_, C1, rows, ns, n_src, dz.data_ptr(), dxyz.data_ptr(), off.data_ptr(), inv.data_ptr(),
                      dproj.data_ptr(), dW0.data_ptr(), dW0.stride(0), _ptr(dwx_ws), _stream())
        else:
            _lib.call("rtk_group_points_grad_set", S_, C1, n_src, rows, ns, dz.data_ptr(), idx.data_ptr(), dproj.data_ptr(), _stream())
            dW0[:, :3] = torch.bmm(dz.view(S_, C1, -1), dxyz.view(S_, 3, -1).transpose(1, 2)).sum(0)
        _, _, dfeats, deferred = _pw_backward(ctx.needs_input_grad[9:9 + nfeat], feats, cols, W0, dproj, False, dW=dW0, owner=ctx.owner)
        return (None if deferred else dW0.view_as(w0), None, None, None, None, None, None, None, None) + tuple(dfeats) + tuple(flat)


def sa_chain_supported(layers):
    chans = [layers[0].conv.out_channels] + [l.conv.out_channels for l in layers[1:]]
    return len(layers) >= 2 and all(c in (16, 32, 64) for c in chans) and all(l.conv.bias is None for l in layers)


def sa_chain(feats, w0, idx, dxyz, layers, row_w, count, groups, inv=None):
    """feats: list of (S,C_i,n_src) tensors whose (virtual) channel concatenation is the level's feature tensor; w0 (C1, 3 + C, 1, 1):
    the first layer's weight [offset columns | feature columns] -- the features are projected per POINT by the feature columns (a
    1x1 conv and a gather commute), gathered by idx and the 3-channel offset term added per (centroid, neighbour) pair;
    idx (S,rows,ns) int32 ball-query indices; dxyz (S,3,rows,ns) neighbour offsets; layers: the SharedMLP's
    Conv2d blocks (conv, bn.bn); inv: optional (off, inv) inverse table of idx (rtk_group_inverse_index) for the gather-form
    backward of the first layer.  Returns the max-pooled (S,C_last,rows) output.  Updates every BatchNorm's running statistics."""
    tensors = []
    for i, l in enumerate(layers):
        if i > 0:
            tensors.append(l.conv.weight)
        tensors += [l.bn.bn.weight, l.bn.bn.bias]
    feats = list(feats)
    return _SAChain.apply(w0, idx, dxyz, row_w, float(count), int(groups), tuple(l.bn.bn for l in layers), inv, len(feats), *feats, *tensors)


# ---- cost volume -------------------------------------------------------------------------------------------------------

class _PackJob(ctypes.Structure):      # rtk_pack_job_t (include/rtk_train.h)
    _fields_ = [("src", ctypes.c_void_p), ("src2", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("pitch", ctypes.c_int), ("transpose", ctypes.c_int), ("kind", ctypes.c_int)]


def _pack_weights(specs, device):
    """specs: list of (kind, matrix-or-vector, transpose, second tensor or None): every kernel image of an operator's live weights in
    ONE launch (rtk_pack_weights) into one workspace.  kind 0 = fragment-major MFMA image (fused.pack_layer), 1 = offset-layer image
    of [W(:, :3) | b] (fused.offset_image), 2 = zero-padded vector (fused.pad_bias).  Returns the list of image tensors (views)."""
    c16 = lambda v: (v + 15) // 16 * 16
    sizes = []
    for kind, t, tr, _ in specs:
        if kind == 0:
            r, c = (t.shape[1], t.shape[0]) if tr else (t.shape[0], t.shape[1])
            sizes.append(c16(r) * c16(c))
        elif kind == 1:
            sizes.append(c16(t.shape[0]) * 4)
        else:
            sizes.append(c16(t.shape[0]))
    ws = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    outs = list(torch.split(ws, sizes))
    jobs = (_PackJob * len(specs))()
    keep = []
    for j, ((kind, t, tr, t2), o) in enumerate(zip(specs, outs)):
        t = t.detach()
        if t.dim() == 2 and t.stride(1) != 1:
            t = t.contiguous()
        keep.append(t)
        jobs[j].src, jobs[j].dst, jobs[j].kind, jobs[j].transpose = t.data_ptr(), o.data_ptr(), kind, int(bool(tr))
        if kind == 2:
            jobs[j].rows, jobs[j].cols, jobs[j].pitch = t.shape[0], 1, 1
        else:
            r, c = (t.shape[1], t.shape[0]) if tr else (t.shape[0], t.shape[1])
            jobs[j].rows, jobs[j].cols, jobs[j].pitch = r, (3 if kind == 1 else c), t.stride(0)
        if t2 is not None:
            t2 = t2.detach().contiguous()
            keep.append(t2)
            jobs[j].src2 = t2.data_ptr()
    _lib.call("rtk_pack_weights", len(specs), jobs, _stream())
    return outs, (ws, keep)


# The cost volume's 256 x 256 products run on the split matrix path (csrc/split_mfma.h: two fp16 pieces, three products) at every batch size.  (At B = 1 the
# fp32-input kernels are 1 % faster per step -- 3.09 vs 3.13 ms: half as many, twice as heavy workgroups on a mostly empty chip --, not
# worth a second product path selected by a batch-size threshold.)  CV_SPLIT = False selects the fp32-input MFMA kernels: the
# comparison implementation of the tests (tests/test_train_gpu.py, tools/grad_parity_report.py --fp32-cv).
CV_SPLIT = True
_SPLIT_IMAGE = fused.SPLIT_IMAGE_256      # int16 elements of one layer's split image


def _cv_split(points=None):
    return CV_SPLIT


class _CvWeights:
    """Packed kernel images of the live cost-volume weights (re-packed every step: the weights are being trained): the four 256x256
    layer images W2, W3, W3^T, W2^T, the offset image of Wd, the WeightNet's three layers and Wc^T -- one launch."""

    def __init__(self, wd, w2, b2, w3, b3, wa, ba, wb, bb, wc, bc, backward, split=None):
        dev = w2.device
        self.is_split = split = CV_SPLIT if split is None else split
        L = fused._Layer
        specs = [(0, w2, False, None), (0, w3, False, None)] + ([(0, w3, True, None), (0, w2, True, None)] if backward else [])
        if split:                                 # split images instead: W2 | W3 | W3^T | W2^T (the transposes straight from w3, w2)
            specs = []
            self.split = torch.empty((4 if backward else 2) * _SPLIT_IMAGE, dtype=torch.int16, device=dev)
            self.split_scales = torch.empty(4, dtype=torch.float32, device=dev)      # inverse weight scales, image by image
            self._w = [w.detach().contiguous() for w in (w2, w3)]
            for k, (w, t) in enumerate([(self._w[0], 0), (self._w[1], 0)] + ([(self._w[1], 1), (self._w[0], 1)] if backward else [])):
                _lib.call("rtk_pack_split_layer", 256, 256, w.data_ptr(), t, self.split[k * _SPLIT_IMAGE:].data_ptr(),
                          self.split_scales[k:].data_ptr(), _stream())
        nm = len(specs)
        specs += [(2, b2, False, None), (2, b3, False, None), (1, wd, False, None), (1, wa, False, ba), (0, wb, False, None), (2, bb, False, None),
                  (0, wc, False, None), (2, bc, False, None), (0, wc, True, None)]
        outs, self._keep = _pack_weights(specs, dev)
        self.blob = outs[0] if nm else None       # the layer images are contiguous in the workspace (blob = W2 | W3 | W3^T | W2^T)
        self.bias = outs[nm]                      # b2 | b3
        self.b2, self.b3 = outs[nm], outs[nm + 1]
        self.layers = (L * nm)()
        for i in range(nm):
            self.layers[i].w_packed = outs[i].data_ptr()
            self.layers[i].bias = outs[nm + min(i, 1)].data_ptr()
            self.layers[i].cin16, self.layers[i].cout16, self.layers[i].act = 16, 16, fused.ACT_LEAKY
        self.wd, self.wa, self.wb, self.bb, self.wc, self.bc, self.wct = outs[nm + 2:nm + 9]
        wn = (L * 3)()
        wn[0].w_packed, wn[0].cin16, wn[0].cout16 = self.wa.data_ptr(), 1, 1
        wn[1].w_packed, wn[1].bias, wn[1].cin16, wn[1].cout16 = self.wb.data_ptr(), self.bb.data_ptr(), 1, 1
        wn[2].w_packed, wn[2].bias, wn[2].cin16, wn[2].cout16 = self.wc.data_ptr(), self.bc.data_ptr(), 1, 16
        self.wn = wn
        if backward and nm:
            self.layers_t = ctypes.cast(ctypes.byref(self.layers, 2 * ctypes.sizeof(L)), ctypes.POINTER(L))      # W3^T, W2^T


class _TnJob(ctypes.Structure):          # rtk_tn_job_t (include/rtk_train.h)
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_pitch", ctypes.c_int),
                ("x_amax", ctypes.c_void_p), ("y_amax", ctypes.c_void_p)]


_lib.SIGNATURES.update({"rtk_tn_gemm256_split": [_i, ctypes.POINTER(_TnJob), ctypes.c_long, _p, ctypes.c_long, _p],
                        "rtk_absmax": [_p, ctypes.c_long, _p, _p]})


TN_MAX_ROWS = (1 << 22) - 16      # rtk_tn_gemm256_split addresses its operands through 32-bit buffer resources: m < 2^22 rows per launch


def tn_gemm256(pairs, amax=None):
    """[(x, y), ...] with x, y (m, 256) fp32 contiguous -> (len(pairs), 256, 256): x^T y of every pair in one launch on the
    split matrix path (rtk_tn_gemm256_split: two fp16 pieces per operand under ONE power-of-two scale per tensor).  amax: [(ax, ay), ...]
    one-element CUDA float tensors >= the operands' largest |element| (the cost-volume kernels hand them over); None: rtk_absmax passes.
    More than TN_MAX_ROWS rows (B * N1 * 16 positions: B = 64 at N = 4096) go in row chunks whose products are added."""
    n, m = len(pairs), pairs[0][0].shape[0]
    dev = pairs[0][0].device
    for x, y in pairs:
        assert x.shape == (m, 256) and y.shape == (m, 256) and x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float32
    if amax is None:
        buf = torch.zeros(2 * n, dtype=torch.float32, device=dev)
        for k, (x, y) in enumerate(pairs):
            _lib.call("rtk_absmax", x.data_ptr(), x.numel(), buf[2 * k:].data_ptr(), _stream())
            _lib.call("rtk_absmax", y.data_ptr(), y.numel(), buf[2 * k + 1:].data_ptr(), _stream())
        amax = [(buf[2 * k:2 * k + 1], buf[2 * k + 1:2 * k + 2]) for k in range(n)]
    total = None
    for r0 in range(0, m, TN_MAX_ROWS):
        mc = min(TN_MAX_ROWS, m - r0)
        out = torch.empty(n, 256, 256, dtype=torch.float32, device=dev)
        jobs = (_TnJob * n)()
        for k, (x, y) in enumerate(pairs):
            jobs[k].x, jobs[k].y, jobs[k].out, jobs[k].out_pitch = x[r0:].data_ptr(), y[r0:].data_ptr(), out[k].data_ptr(), 256
            jobs[k].x_amax, jobs[k].y_amax = amax[k][0].data_ptr(), amax[k][1].data_ptr()
        steps = (mc + 15) // 16
        slabs = max(1, min(256 // n, (steps + 7) // 8))
        ws = torch.empty(n * slabs * 65536, dtype=torch.float32, device=dev)
        _lib.call("rtk_tn_gemm256_split", n, jobs, mc, ws.data_ptr(), ws.numel(), _stream())
        total = out if total is None else total.add_(out)
    return total


class _CostVolume(torch.autograd.Function):
    """out[i] = sum_k WeightNet(d_ik) * mlp(p1[i] + p2[knn[i,k]] + Wd d_ik),  d_ik = xyz2[knn[i,k]] - xyz1[i]
    (utils/model_utils/model_utils.py:216-236 with the first conv split by input segment).  Forward = the inference
    kernel rtk_cost_volume; backward = rtk_cost_volume_bwd + rtk_tn_gemm256_split for the weight gradients."""

    @staticmethod
    def forward(ctx, p1, p2, wd, w2, b2, w3, b3, wa, ba, wb, bb, wc, bc, xyz1, xyz2, knn):
        B, n1, _ = xyz1.shape
        n2 = xyz2.shape[1]
        p1, p2 = p1.contiguous(), p2.contiguous()
        # all kernel images are built once per step, here: the backward reuses them (ctx.images)
        W = _CvWeights(wd, w2, b2, w3, b3, wa, ba, wb, bb, wc, bc, backward=True, split=_cv_split(B * n1))
        ctx.images = W
        out = torch.empty(B * n1, 256, dtype=torch.float32, device=p1.device)
        # the three activations are kept for the backward (3 x 268 MB at B = 64 of 288 GB): it then needs no recomputation and its
        # weight-gradient GEMMs read the same tensors
        acts = torch.empty(3, B * n1 * 16, 256, dtype=torch.float32, device=p1.device)
        masks = torch.empty(2, B * n1 * 16, 4, dtype=torch.int64, device=p1.device)          # sign bits of a1, a2 (kernel lane order)
        # the largest |element| of a1, a2 (forward) and dz3, dz2 (backward), folded in by the kernels: the weight-gradient contraction
        # over the positions takes ONE power-of-two scale per tensor (rtk_tn_gemm256_split)
        ctx.amax = amax = arena_zeros(4, torch.float32, p1.device)
        if W.is_split:
            _lib.call("rtk_cost_volume_split_train", B, n1, n2, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), p1.data_ptr(), p2.data_ptr(),
                      W.wd.data_ptr(), W.split.data_ptr(), W.split_scales.data_ptr(), W.b2.data_ptr(), W.b3.data_ptr(), W.wn, out.data_ptr(), 256,
                      acts[0].data_ptr(),
                      acts[1].data_ptr(), acts[2].data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), amax.data_ptr(), _stream())
        else:
            _lib.call("rtk_cost_volume_train", B, n1, n2, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), p1.data_ptr(), p2.data_ptr(),
                      W.wd.data_ptr(), W.layers, W.wn, out.data_ptr(), 256, acts[0].data_ptr(), acts[1].data_ptr(), acts[2].data_ptr(),
                      masks[0].data_ptr(), masks[1].data_ptr(), _stream())
        ctx.save_for_backward(acts, masks, wa, ba, wb, bb, wc, xyz1, xyz2, knn)
        return out

    @staticmethod
    def backward(ctx, dout):
        acts, masks, wa, ba, wb, bb, wc, xyz1, xyz2, knn = ctx.saved_tensors
        B, n1, _ = xyz1.shape
        n2 = xyz2.shape[1]
        dev = acts.device
        M = B * n1 * 16
        dout = dout.contiguous()
        W = ctx.images
        a1, a2, a3 = acts.unbind(0)
        big = torch.empty(4, M, 256, dtype=torch.float32, device=dev)
        dz1, dz2, dz3, dq3 = big.unbind(0)
        d4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        dt2 = torch.empty(M, 8, dtype=torch.float32, device=dev)
        dp1 = torch.empty(B * n1, 256, dtype=torch.float32, device=dev)
        dpd = torch.empty(B * n1, 3, 256, dtype=torch.float32, device=dev)
        dbr = torch.empty(B * n1, 512, dtype=torch.float32, device=dev)       # per-query neighbour sums of dz3 | dz2
        if W.is_split:
            _lib.call("rtk_cost_volume_bwd_split", B, n1, n2, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), W.split[2 * _SPLIT_IMAGE:].data_ptr(),
                      W.split_scales[2:].data_ptr(), W.wn, dout.data_ptr(), 256, a3.data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), dz1.data_ptr(), dz2.data_ptr(),
                      dz3.data_ptr(), dq3.data_ptr(), d4.data_ptr(), dp1.data_ptr(), dpd.data_ptr(), dt2.data_ptr(), dbr.data_ptr(),
                      ctx.amax[2:].data_ptr(), _stream())
        else:
            _lib.call("rtk_cost_volume_bwd", B, n1, n2, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), W.layers_t, W.wn, W.wct.data_ptr(),
                      dout.data_ptr(), 256, a3.data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), dz1.data_ptr(), dz2.data_ptr(), dz3.data_ptr(),
                      dq3.data_ptr(), d4.data_ptr(), dp1.data_ptr(), dpd.data_ptr(), dt2.data_ptr(), dbr.data_ptr(), _stream())
        dp2 = torch.empty(B * n2, 256, dtype=torch.float32, device=dev)
        _lib.call("rtk_scatter_add_rows", B, n1 * 16, n2, 256, knn.data_ptr(), dz1.data_ptr(), dp2.data_ptr(), _stream())
        # weight gradients: contractions over the M positions, dW2 = dz2^T a1 and dW3 = dz3^T a2, in one launch on the split-bf16
        # matrix path (rtk_tn_gemm256_split; until round 3 a batched library GEMM on the fp32 pipe)
        am = ctx.amax if W.is_split else None        # (a1, a2, dz3, dz2); the fp32-input comparison kernels emit none: rtk_absmax passes
        dw2, dw3 = tn_gemm256([(dz2, a1), (dz3, a2)], None if am is None else [(am[3:4], am[0:1]), (am[2:3], am[1:2])]).unbind(0)
        db3, db2 = dbr.sum(0).split(256)
        dwd = dpd.sum(0).t()
        dwa, dba, dwb, dbb, dwc, dbc = _weightnet_backward(d4, dq3, dt2, wa, ba, wb, bb, wc)
        return dp1, dp2, dwd, dw2, db2, dw3, db3, dwa, dba, dwb, dbb, dwc, dbc, None, None, None


def cost_volume(p1, p2, wd, w2, b2, w3, b3, wa, ba, wb, bb, wc, bc, xyz1, xyz2, knn):
    """p1 (B*n1,256) (first-layer bias included), p2 (B*n2,256), wd (256,3), w2/w3 (256,256), b2/b3 (256),
    WeightNet wa (8,3) ba (8) wb (8,8) bb (8) wc (256,8) bc (256), xyz1 (B,n1,3), xyz2 (B,n2,3), knn (B,n1,16) int64
    -> (B*n1, 256)."""
    assert xyz1.is_contiguous() and xyz2.is_contiguous() and knn.is_contiguous() and knn.dtype == torch.int64
    return _CostVolume.apply(p1, p2, wd, w2, b2, w3, b3, wa, ba, wb, bb, wc, bc, xyz1, xyz2, knn)


def time_cost_volume_bwd(batch, n, dev, iters=10):
    """Measurement helper (bench.py): rtk_cost_volume_bwd at the bench shape on random operands, `iters` back-to-back launches
    between two HIP events on the current stream.  Returns (ms per launch, FLOPs per launch): per (point, neighbour) pair the
    two 256x256 input gradients, Wc^T dq3 and the WeightNet recomputation + masks."""
    g = torch.Generator(dev).manual_seed(0)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    B, M = batch, batch * n * 16
    xyz1, xyz2 = r(B, n, 3).contiguous(), r(B, n, 3).contiguous()
    knn = torch.randint(0, n, (B, n, 16), device=dev, generator=g)
    dout = r(B * n, 256)
    w2, w3 = r(256, 256) * 0.06, r(256, 256) * 0.06
    W = _CvWeights(r(256, 3), w2, r(256), w3, r(256), r(8, 3), r(8), r(8, 8), r(8), r(256, 8), r(256), backward=True, split=_cv_split(B * n))
    acts = r(3, M, 256)
    masks = torch.randint(-2 ** 62, 2 ** 62, (2, M, 4), device=dev, generator=g)
    big = torch.empty(4, M, 256, device=dev)
    d4, dt2 = torch.empty(M, 4, device=dev), torch.empty(M, 8, device=dev)
    dp1, dpd = torch.empty(B * n, 256, device=dev), torch.empty(B * n, 3, 256, device=dev)
    dbr = torch.empty(B * n, 512, device=dev)
    st = _stream()

    def launch():
        if W.is_split:
            _lib.call("rtk_cost_volume_bwd_split", B, n, n, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), W.split[2 * _SPLIT_IMAGE:].data_ptr(),
                      W.split_scales[2:].data_ptr(), W.wn, dout.data_ptr(), 256, acts[2].data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), big[0].data_ptr(),
                      big[1].data_ptr(), big[2].data_ptr(), big[3].data_ptr(), d4.data_ptr(), dp1.data_ptr(), dpd.data_ptr(), dt2.data_ptr(),
                      dbr.data_ptr(), None, st)
            return
        _lib.call("rtk_cost_volume_bwd", B, n, n, xyz1.data_ptr(), xyz2.data_ptr(), knn.data_ptr(), W.layers_t, W.wn, W.wct.data_ptr(),
                  dout.data_ptr(), 256, acts[2].data_ptr(), masks[0].data_ptr(), masks[1].data_ptr(), big[0].data_ptr(), big[1].data_ptr(),
                  big[2].data_ptr(), big[3].data_ptr(), d4.data_ptr(), dp1.data_ptr(), dpd.data_ptr(), dt2.data_ptr(), dbr.data_ptr(), st)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * (2 * 256 * 256 + (3 * 8 + 8 * 8 + 8 * 256) + 256 + 8 * 256)
    return ms, flops


def _weightnet_images(wa, ba, wb, bb, wc, bc):
    """Kernel images of a WeightNet 3 -> 8 -> 8 -> 256 (see fused._WeightNet) from live parameters, + Wc^T; one launch."""
    L = fused._Layer
    outs, keep = _pack_weights([(1, wa, False, ba), (0, wb, False, None), (2, bb, False, None), (0, wc, False, None), (2, bc, False, None),
                                (0, wc, True, None)], wa.device)
    wn = (L * 3)()
    wn[0].w_packed, wn[0].cin16, wn[0].cout16 = outs[0].data_ptr(), 1, 1
    wn[1].w_packed, wn[1].bias, wn[1].cin16, wn[1].cout16 = outs[1].data_ptr(), outs[2].data_ptr(), 1, 1
    wn[2].w_packed, wn[2].bias, wn[2].cin16, wn[2].cout16 = outs[3].data_ptr(), outs[4].data_ptr(), 1, 16
    return wn, (outs, keep)


def _weightnet_backward(d4, dq3, dt2, wa, ba, wb, bb, wc):
    """Gradients of the WeightNet parameters from dq3 (M,256) (last pre-activation), dt2 = dq3 Wc (M,8) and the direction vectors
    d4[:, :3]: one kernel (rtk_weightnet_bwd) recomputes the (M,8) hidden activations and accumulates all six gradients."""
    M, C = dq3.shape
    dev = d4.device
    buf = _zeros((24 + 8 + 64 + 8 + C * 8 + C,), torch.float32, dev)
    dwa, dba, dwb, dbb, dwc, dbc = torch.split(buf, [24, 8, 64, 8, C * 8, C])
    c = lambda t: t.detach().contiguous()
    wa_, ba_, wb_, bb_ = c(wa), c(ba), c(wb), c(bb)
    ws = torch.empty(1024 * ((9 * C + 107) & ~3), dtype=torch.float32, device=dev)          # workgroup partials (uninitialised scratch)
    _lib.call("rtk_weightnet_bwd", M, C, d4.data_ptr(), dq3.data_ptr(), dt2.data_ptr(), wa_.data_ptr(), ba_.data_ptr(), wb_.data_ptr(),
              bb_.data_ptr(), dwa.data_ptr(), dba.data_ptr(), dwb.data_ptr(), dbb.data_ptr(), dwc.data_ptr(), dbc.data_ptr(), ws.data_ptr(),
              ws.numel(), _stream())
    return dwa.view(8, 3), dba, dwb.view(8, 8), dbb, dwc.view(C, 8), dbc


class _PatchCost(torch.autograd.Function):
    """out[i] = sum_k WeightNet(xyz[knn[i,k]] - xyz[i]) * feat[knn[i,k]]  (model_utils.py:238-248): forward = the inference
    kernel rtk_patch_cost, backward = rtk_patch_cost_bwd + the LDS scatter + the WeightNet's small GEMMs."""

    @staticmethod
    def forward(ctx, feat, wa, ba, wb, bb, wc, bc, xyz, knn, live):
        B, n, _ = xyz.shape
        feat = feat.contiguous()
        wn, keep = _weightnet_images(wa, ba, wb, bb, wc, bc)
        ctx.images = (wn, keep, keep[0][5])                         # reused by the backward (keep[0][5] = packed Wc^T)
        ctx.live = live                                             # (B,) int32 or None: query points past it are padding copies of point 0
        out = torch.empty(B * n, 256, dtype=torch.float32, device=feat.device)
        _lib.call("rtk_patch_cost", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, out.data_ptr(), 256, 0, _stream())
        ctx.save_for_backward(feat, wa, ba, wb, bb, wc, bc, xyz, knn)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, wa, ba, wb, bb, wc, bc, xyz, knn = ctx.saved_tensors
        B, n, _ = xyz.shape
        M = B * n * 16
        dev = feat.device
        dout = dout.contiguous()
        wn, keep, wct = ctx.images
        dq3 = torch.empty(M, 256, dtype=torch.float32, device=dev)
        dt2 = torch.empty(M, 8, dtype=torch.float32, device=dev)
        d4 = torch.empty(M, 4, dtype=torch.float32, device=dev)
        dfeat = torch.empty(B * n, 256, dtype=torch.float32, device=dev)
        if n <= 2048:                          # the 16-bit inverse table and its LDS budget (rtk_group_inverse_index)
            # feature gradient as a gather over the inverse kNN table (positions sorted by the row they gathered), the WeightNet output
            # recomputed from its hidden activation: nothing of size (M, 256) is materialised for it
            t2 = torch.empty(M, 8, dtype=torch.float32, device=dev)
            _lib.call("rtk_patch_cost_bwd", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, wct.data_ptr(), dout.data_ptr(),
                      256, None, dq3.data_ptr(), dt2.data_ptr(), d4.data_ptr(), t2.data_ptr(), _stream())
            k32 = knn.to(torch.int32)
            off = torch.empty(B, n + 1, dtype=torch.int32, device=dev)
            inv = torch.empty(B, 16 * n, dtype=torch.int16, device=dev)
            # padded clouds: the padding queries (copies of point 0) stay out of the table, their gradient rows are added to row 0's
            group_inverse_index_multi(B, [(n, 16 * n, k32, off, inv, ctx.live, 16)])
            wc_, bc_ = wc.detach().reshape(256, 8).contiguous(), bc.detach().contiguous()
            row0 = torch.empty(B, 256, dtype=torch.float32, device=dev) if ctx.live is not None else None
            _lib.call("rtk_patch_dfeat_gather", B, n, off.data_ptr(), inv.data_ptr(), t2.data_ptr(), wc_.data_ptr(), bc_.data_ptr(),
                      dout.data_ptr(), 256, dfeat.data_ptr(), _ptr(ctx.live), _ptr(row0), _stream())
        else:
            dxg = torch.empty(M, 256, dtype=torch.float32, device=dev)
            _lib.call("rtk_patch_cost_bwd", B, n, xyz.data_ptr(), knn.data_ptr(), feat.data_ptr(), 256, wn, wct.data_ptr(), dout.data_ptr(),
                      256, dxg.data_ptr(), dq3.data_ptr(), dt2.data_ptr(), d4.data_ptr(), None, _stream())
            _lib.call("rtk_scatter_add_rows", B, n * 16, n, 256, knn.data_ptr(), dxg.data_ptr(), dfeat.data_ptr(), _stream())
        return (dfeat,) + _weightnet_backward(d4, dq3, dt2, wa, ba, wb, bb, wc) + (None, None, None)


def patch_cost(feat, wa, ba, wb, bb, wc, bc, xyz, knn, live=None):
    """feat (B*n,256) point-major, WeightNet parameters as in cost_volume, xyz (B,n,3), knn (B,n,16) int64 -> (B*n,256).
    live (B,) int32 on the device: padded clouds -- the query points from live[b] on are copies of point 0 (the backward folds their
    gradient into point 0's instead of walking their entries in the inverse table)."""
    assert xyz.is_contiguous() and knn.is_contiguous() and knn.dtype == torch.int64
    return _PatchCost.apply(feat, wa, ba, wb, bb, wc, bc, xyz, knn, live)


# ---- 1x1 convolution with a GEMM weight gradient ----------------------------------------------------------------------

class _Conv1x1(torch.autograd.Function):
    """z = W x for x (S,Cin,rows,ns) NCHW, W (Cout,Cin,1,1).  Forward and input gradient are the framework's convolution;
    the WEIGHT gradient dW = sum_b dz_b x_b^T contracts over the positions, which are the contiguous axis of both NCHW
    operands -- a batched NT GEMM as is.  (MIOpen's backward-weights path first transposes both tensors to NHWC, which
    costs more than the contraction for these narrow layers.)"""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.conv2d(x, w)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        S_, Co = dz.shape[:2]
        Ci = x.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv2d_input(x.shape, w, dz)      # (a broadcast GEMM W^T dz_b was measured slower)
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dz.view(S_, Co, -1), x.reshape(S_, Ci, -1).transpose(1, 2)).sum(0).view_as(w)
        return dx, dw


def conv1x1(x, w):
    return _Conv1x1.apply(x, w)


# ---- GRU step -------------------------------------------------------------------------------------------------------------

class _GRUStep(torch.autograd.Function):
    """nn.GRU(H, H, L) on a length-1 sequence (model_utils.py:279,296): forward = the inference kernel rtk_gru_step, backward
    = rtk_gru_step_bwd + rtk_gru_wgrad for the parameter gradients (MIOpen's RNN issues ~150 kernels for this 64 x 128
    problem).  params = (w_ih_l0, w_hh_l0, b_ih_l0, b_hh_l0, w_ih_l1, ...)."""

    @staticmethod
    def forward(ctx, x, h_in, *params):
        ctx.set_materialize_grads(False)      # (h_out usually feeds nothing: no zero tensor for its gradient)
        L = len(params) // 4
        B, H = x.shape
        x, h_in = x.contiguous(), h_in.contiguous()
        # the kernels' stacked weight images (plain and transposed) from the live per-layer parameters: one launch
        buf = torch.empty(4 * L * 3 * H * H + 2 * L * 3 * H, dtype=torch.float32, device=x.device)
        n = L * 3 * H * H
        w_ih, w_ih_t, w_hh, w_hh_t = buf[:n].view(L, 3 * H, H), buf[n:2 * n].view(L, H, 3 * H), buf[2 * n:3 * n].view(L, 3 * H, H), buf[3 * n:4 * n].view(L, H, 3 * H)
        b_ih, b_hh = buf[4 * n:4 * n + L * 3 * H].view(L, 3 * H), buf[4 * n + L * 3 * H:].view(L, 3 * H)
        ptrs = (ctypes.c_void_p * (4 * L))(*[p.detach().contiguous().data_ptr() for p in params])
        _lib.call("rtk_gru_pack_params", L, H, ptrs, w_ih.data_ptr(), w_ih_t.data_ptr(), w_hh.data_ptr(), w_hh_t.data_ptr(), b_ih.data_ptr(),
                  b_hh.data_ptr(), _stream())
        h_out = torch.empty(L, B, H, dtype=torch.float32, device=x.device)
        y = torch.empty(B, H, dtype=torch.float32, device=x.device)
        _lib.call("rtk_gru_step", B, L, H, x.data_ptr(), h_in.data_ptr(), w_ih_t.data_ptr(), w_hh_t.data_ptr(), b_ih.data_ptr(),
                  b_hh.data_ptr(), h_out.data_ptr(), y.data_ptr(), _stream())
        ctx.save_for_backward(x, h_in, h_out, w_ih, w_hh, w_ih_t, w_hh_t, b_ih, b_hh)
        return y, h_out

    @staticmethod
    def backward(ctx, dy, dh_out):
        x, h_in, h_out, w_ih, w_hh, w_ih_t, w_hh_t, b_ih, b_hh = ctx.saved_tensors
        L, B, H = h_out.shape
        dev = x.device
        dy = dy.contiguous() if dy is not None else torch.zeros(B, H, dtype=torch.float32, device=dev)
        dho = dh_out.contiguous() if dh_out is not None else None
        dx = torch.empty(B, H, dtype=torch.float32, device=dev)
        dh_in = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        dg = torch.empty(2, L, B, 3 * H, dtype=torch.float32, device=dev)
        _lib.call("rtk_gru_step_bwd", B, L, H, x.data_ptr(), h_in.data_ptr(), h_out.data_ptr(), w_ih_t.data_ptr(), w_hh_t.data_ptr(),
                  w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), dy.data_ptr(), _ptr(dho), dx.data_ptr(),
                  dh_in.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), _stream())
        dw = torch.empty(2, L, 3 * H, H, dtype=torch.float32, device=dev)                 # dW_ih | dW_hh
        db = torch.empty(2, L, 3 * H, dtype=torch.float32, device=dev)
        _lib.call("rtk_gru_wgrad", B, L, H, x.data_ptr(), h_in.data_ptr(), h_out.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), dw[0].data_ptr(),
                  dw[1].data_ptr(), db[0].data_ptr(), db[1].data_ptr(), _stream())
        grads = []
        for l in range(L):
            grads += [dw[0, l], dw[1, l], db[0, l], db[1, l]]
        return (dx, dh_in) + tuple(grads)


def gru_step(x, h_in, gru):
    """x (B,H), h_in (L,B,H), gru: nn.GRU(H, H, L) -> (y (B,H), h_out (L,B,H))."""
    L = gru.num_layers
    assert gru.input_size == gru.hidden_size == 128 and not gru.bidirectional and gru.bias and not gru.batch_first
    params = []
    for l in range(L):
        params += [getattr(gru, "weight_ih_l%d" % l), getattr(gru, "weight_hh_l%d" % l), getattr(gru, "bias_ih_l%d" % l),
                   getattr(gru, "bias_hh_l%d" % l)]
    return _GRUStep.apply(x, h_in, *params)


# ---- three-NN interpolation with a gather-form backward --------------------------------------------------------------------------

class _ThreeInterpolate(torch.autograd.Function):
    """pointnet2_utils.three_interpolate (features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)) whose backward reads the
    inverse table of idx (off (B,M+1), inv (B,3n), rtk_group_inverse_index) instead of scattering with LDS float atomics."""

    @staticmethod
    def forward(ctx, features, idx, weight, off, inv, event, n_valid=None):
        from . import pointnet2_hip as _native
        features = features.contiguous()
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        _native.three_interpolate_wrapper(B, c, m, n, features, idx, weight, out)
        ctx.save_for_backward(weight, off, inv)
        ctx.m, ctx.event, ctx.n_valid = m, event, n_valid
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, off, inv = ctx.saved_tensors
        B, c, n = grad_out.shape
        if ctx.event is not None:                                  # table built on the geometry stream (TrainGeometry)
            torch.cuda.current_stream().wait_event(ctx.event)
        grad = torch.empty((B, c, ctx.m), dtype=torch.float32, device=grad_out.device)
        _lib.call("rtk_three_interpolate_grad_gather", B, c, n, ctx.m, grad_out.contiguous().data_ptr(), weight.data_ptr(), off.data_ptr(),
                  inv.data_ptr(), grad.data_ptr(), _ptr(ctx.n_valid), _stream())
        return grad, None, None, None, None, None, None


def three_interpolate(features, idx, weight, inv_table, event=None, n_valid=None):
    """inv_table = (off, inv) of idx.view(B, 3n) over the M known points.  n_valid (B,) int32: padded clouds whose table was built
    without the padding points' positions (their gradient is folded into point 0's, rtk_three_interpolate_grad_gather)."""
    return _ThreeInterpolate.apply(features, idx, weight, inv_table[0], inv_table[1], event, n_valid)


# ---- multi-task loss -------------------------------------------------------------------------------------------------------------

class _BackboneLoss(torch.autograd.Function):
    """loss.backbone_loss (values and gradients) as one kernel: -> (Loss, items (4) = [Loss, SceneFlowLoss, TrackingLoss, SegLoss])."""

    @staticmethod
    def forward(ctx, flow, cls, pc1, gt_warp, gt_cls, pretrain, n_valid=None):
        B, _, N = pc1.shape
        dev = pc1.device
        flow, cls, pc1, gt_warp = flow.contiguous(), cls.contiguous(), pc1.contiguous(), gt_warp.contiguous()
        g = gt_cls.to(torch.uint8) if gt_cls.dtype != torch.bool else gt_cls.view(torch.uint8)
        g = g.contiguous()
        stride = 0 if g.dim() == 1 else N
        # NOT from the zero arena: the caller keeps the items across steps (Trainer.step returns them without a host sync)
        items = torch.zeros(5 + 2 * B, dtype=torch.float32, device=dev)      # the four items | arrival counter | the samples' shares
        dflow = None if pretrain else torch.empty(B, 3, N, dtype=torch.float32, device=dev)
        dcls = torch.empty(B, N, dtype=torch.float32, device=dev)
        _lib.call("rtk_backbone_loss", B, N, pc1.data_ptr(), flow.data_ptr(), gt_warp.data_ptr(), cls.data_ptr(), g.data_ptr(), stride,
                  int(bool(pretrain)), items.data_ptr(), _ptr(dflow), dcls.data_ptr(), _ptr(n_valid), _stream())
        ctx.save_for_backward(dflow, dcls)
        items = items[:4]
        ctx.mark_non_differentiable(items)
        return items[0], items                           # the differentiable total as its own output: no select_backward (zeros + copy)

    @staticmethod
    def backward(ctx, g, _):
        dflow, dcls = ctx.saved_tensors
        return (None if dflow is None else dflow * g), dcls * g, None, None, None, None, None


def backbone_loss(pc1, flow, cls, gt_warp, gt_cls, pretrain=False, n_valid=None):
    """(total, items dict) of loss.backbone_loss(pc1 + flow, cls, gt_warp, gt_cls, pretrain) with total = items['Loss'];
    CUDA fp32 only.  n_valid (B,) int32 on the device: padded batch -- sample b's loss is that of its first n_valid[b] points."""
    if n_valid is not None:
        assert n_valid.is_cuda and n_valid.dtype == torch.int32 and n_valid.is_contiguous() and n_valid.numel() == pc1.shape[0]
    total, it = _BackboneLoss.apply(flow, cls, pc1, gt_warp, gt_cls, bool(pretrain), n_valid)
    return total, {"Loss": total, "SceneFlowLoss": it[1], "TrackingLoss": it[2], "SegLoss": it[3]}
