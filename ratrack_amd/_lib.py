"""ctypes binding of librtk_hip.so (the C ABI declared in include/rtk_pointnet2.h).

There is NO CPU or eager fallback: if the HIP library is missing or fails to load, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

SO_PATH = os.path.join(_HERE, "lib", "librtk_hip.so")

_c_int, _c_float, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (restype is always int); mirrors include/rtk_pointnet2.h one to one
SIGNATURES = {
    "rtk_furthest_point_sampling": [_c_int] * 3 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_gather_points": [_c_int] * 4 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_gather_points_grad": [_c_int] * 4 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_ball_query": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 3 + [_c_void_p],
    "rtk_group_points": [_c_int] * 5 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_group_points_grad": [_c_int] * 5 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_group_points_grad_set": [_c_int] * 5 + [_c_void_p] * 3 + [_c_void_p],
    "rtk_three_nn": [_c_int] * 3 + [_c_void_p] * 4 + [_c_void_p],
    "rtk_knn": [_c_int] * 4 + [_c_void_p] * 4 + [_c_void_p],
    "rtk_three_interpolate": [_c_int] * 4 + [_c_void_p] * 4 + [_c_void_p],
    "rtk_three_interpolate_grad": [_c_int] * 4 + [_c_void_p] * 4 + [_c_void_p],
    "rtk_three_interpolate_grad_set": [_c_int] * 4 + [_c_void_p] * 4 + [_c_void_p],
    "rtk_knn_point": [_c_int] * 4 + [_c_void_p] * 3 + [_c_void_p],
}

_lib = None


class RtkError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises RtkError if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        # not a fallback: build the product itself (hipcc for gfx950) if this checkout has never been built
        try:
            from . import build as _build
            _build.build(verbose=False)
        except Exception as e:
            raise RtkError("librtk_hip.so not found at %s and building it failed (%s) -- run `python -m ratrack_amd.build` "
                           "(there is no CPU fallback)" % (SO_PATH, e))
    try:
        lib = ctypes.CDLL(SO_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RtkError("cannot load %s: %s" % (SO_PATH, e))
    lib.rtk_last_error.restype = ctypes.c_char_p
    lib.rtk_version.restype = _c_int
    _lib = lib
    return lib


_bound = {}


def _fn(name):
    fn = _bound.get(name)
    if fn is None:
        fn = getattr(load(), name)        # AttributeError if the library does not export it
        fn.argtypes = SIGNATURES[name]
        fn.restype = _c_int
        _bound[name] = fn
    return fn


TIMING = None      # a list: every call appends (entry point, start event, stop event) on the current stream (bench.py, one eager pass)


def call(name, *args):
    lib = load()
    if TIMING is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _fn(name)(*args)
        e1.record()
        TIMING.append((name, e0, e1))
    else:
        rc = _fn(name)(*args)
    if rc != 0:
        raise RtkError("%s failed (%d): %s" % (name, rc, lib.rtk_last_error().decode()))
    return rc
