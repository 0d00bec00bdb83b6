"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import glob
import os
import re

import pytest

from ratrack_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        names += re.findall(r"RTK_EXPORT\s+[\w\s\*]+?\b(rtk_\w+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_reference_surface():
    names = declared_symbols()
    # one entry per pybind export of the reference (pointnet2_api.cpp:10-25) + knn_point
    for n in ["rtk_ball_query", "rtk_group_points", "rtk_group_points_grad", "rtk_gather_points",
              "rtk_gather_points_grad", "rtk_furthest_point_sampling", "rtk_knn", "rtk_three_nn",
              "rtk_three_interpolate", "rtk_three_interpolate_grad", "rtk_knn_point"]:
        assert n in names


def test_library_builds_loads_and_exports_everything():
    so = build.build(verbose=False)
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for name in declared_symbols():
        assert hasattr(lib, name), "librtk_hip.so does not export %s" % name
    lib.rtk_version.restype = ctypes.c_int
    assert lib.rtk_version() >= 1
    # the Python binding knows every compute entry point the headers declare (fused.py registers rtk_fused.h's)
    import ratrack_amd.fused  # noqa: F401
    import ratrack_amd.train_ops  # noqa: F401  (registers rtk_train.h)
    import ratrack_amd.optim  # noqa: F401  (rtk_adam_multi)
    bound = set(_lib.SIGNATURES) | {"rtk_last_error", "rtk_version"}
    assert set(declared_symbols()) <= bound, set(declared_symbols()) - bound


def test_no_cpu_fallback_in_product():
    """The product path must not import, load or execute anything under oracle/ (nor any CPU fallback):
    no import of the package, no string constant naming it outside docstrings."""
    import ast
    pkg = os.path.join(ROOT, "ratrack_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        tree = ast.parse(open(path).read())
        docstrings = set()
        for node in ast.walk(tree):
            if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)) and node.body and \
                    isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], "value", None), ast.Constant):
                docstrings.add(id(node.body[0].value))
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                assert not any(a.name.split(".")[0] == "oracle" for a in node.names), path
            elif isinstance(node, ast.ImportFrom):
                assert (node.module or "").split(".")[0] != "oracle", path
            elif isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in docstrings:
                assert "oracle" not in node.value, "%s: string constant mentions oracle: %r" % (path, node.value[:60])
    # and the native library does not link the oracle
    for src in glob.glob(os.path.join(pkg, "csrc", "*")):
        code = re.sub(r"/\*.*?\*/", "", open(src).read(), flags=re.S)              # comments may cite the oracle
        code = re.sub(r"//[^\n]*", "", code)
        assert "__global__" in code or not src.endswith(".hip")
        assert "pointnet2_ref" not in code and "rtk_ref_" not in code, src


def test_ops_refuse_cpu_tensors():
    import torch
    from ratrack_amd import pointnet2_hip
    with pytest.raises(_lib.RtkError):
        pointnet2_hip.ball_query_wrapper(1, 4, 2, 1.0, 2, torch.zeros(1, 2, 3), torch.zeros(1, 4, 3),
                                         torch.zeros(1, 2, 2, dtype=torch.int32))
