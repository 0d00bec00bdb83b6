"""Shared helpers for the parity tests (golden fixtures, deterministic weights, tolerances)."""
import json
import os

import numpy as np
import torch

from ratrack_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

EVAL_CASES = ["eval_b2_n256", "eval_b1_n242", "eval_b1_n1024", "eval_b1_n256_dups"]
# north_star tolerance: "within 1e-4 rel for scene-flow floats", defined per tensor as
# max|a-b| <= RTOL * max|b| (flows are ~0 for static points, element-wise relative
# error is meaningless there).
RTOL = 1e-4


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def state_dict_spec():
    with open(os.path.join(GOLDEN, "state_dict_spec.json")) as f:
        return json.load(f)


def reference_state_dict(device="cpu"):
    """The reference's 361-entry state dict, values from the deterministic generator."""
    spec = state_dict_spec()["entries"]
    sd = {}
    for k, (shape, dtype) in spec.items():
        a = synth.tensor_for_key(k, tuple(shape), dtype_is_int=(dtype == "int64"))
        sd[k] = torch.from_numpy(np.ascontiguousarray(a)).reshape(shape).to(device)
    return sd


def inputs_of(case, device="cpu"):
    g = lambda k: torch.from_numpy(case["in_" + k]).to(device)
    return g("pc1"), g("pc2"), g("feature1"), g("feature2")


def rel_err(a, b):
    """max|a-b| / max|b| with a, b array-likes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, rtol=RTOL, what=""):
    e = rel_err(a, b)
    assert e <= rtol, "%s: rel-to-scale error %.3e > %.1e" % (what, e, rtol)


# ---- gradient records of the round-3 fixtures (tools/make_golden.py: grad_records) ----------------------------------------
REAL_CASES = ["real_549_1047", "real_1047_1201", "real_1201_549"]
GRAD_SAMPLE = 1024


def grad_stride(numel):
    s = -(-numel // GRAD_SAMPLE)
    return s | 1 if s > 1 else 1


def probe_vector(key, numel):
    return 2.0 * synth._uniform01("probe/" + key, numel, 77) - 1.0


def grad_sample(g):
    a = np.asarray(g, dtype=np.float64).ravel()
    return a[::grad_stride(a.size)]


def grad_report(case, grads, prefix=""):
    """grads: {parameter name: array or None}.  Compares with the fixture's gradient records (`prefix`grad/..., grad_norms,
    grad_probes, arb_*).  Returns a list of rows: e_ref = max|mine - reference fp32| over the sampled elements / the reference
    tensor's largest element, e_arb the same against the float64 arbiter (None without arbiter samples), floor = the fp32
    oracle's own distance from the arbiter.  Parameters whose EXACT gradient is zero -- a bias in front of a BatchNorm, the GRU
    under B = 1 batch statistics; recognised by a float64 norm below 1e-8 of the model's largest -- carry only rounding noise
    (1e-7 of the model's gradient scale in the reference too): for them the errors are measured against 1e-4 of the model's
    largest gradient element, i.e. `e <= 1` means "noise, as in the reference"."""
    names = [str(k) for k in case[prefix + "grad_names"]]
    norms = case[prefix + "grad_norms"]
    arb_norms = case[prefix + "arb_grad_norms"]
    gmax = max(float(np.abs(case[prefix + "grad/" + k]).max()) for k, n in zip(names, norms) if n >= 0)
    floor = case.get(prefix + "arb_fp32_oracle_relerr")
    rows = []
    for i, (k, n) in enumerate(zip(names, norms)):
        g = grads.get(k)
        if n < 0:
            assert g is None or float(np.abs(np.asarray(g)).max()) == 0.0, "%s: dead parameter has a gradient" % k
            continue
        assert g is not None, "%s: no gradient" % k
        zero = float(arb_norms[i]) <= 1e-8 * float(arb_norms.max())
        mine = grad_sample(g)
        ref = case[prefix + "grad/" + k].astype(np.float64)
        scale = 1e-4 * gmax if zero else float(np.abs(ref).max())
        e_ref = float(np.abs(mine - ref).max() / scale)
        e_arb = None
        ak = prefix + "arb_grad/" + k
        if ak in case:
            arb = case[ak].astype(np.float64)
            e_arb = float(np.abs(mine - arb).max() / (1e-4 * gmax if zero else float(np.abs(arb).max())))
        full = np.asarray(g, dtype=np.float64).ravel()
        probe = float((full * probe_vector(k, full.size)).sum())
        rows.append(dict(name=k, zero=zero, e_ref=e_ref, e_arb=e_arb, floor=None if floor is None or zero else float(floor[i]),
                         norm=float(np.sqrt((full * full).sum())), ref_norm=float(n), probe=probe,
                         ref_probe=float(case[prefix + "grad_probes"][i]), gmax=gmax))
    return rows
