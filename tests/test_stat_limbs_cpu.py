"""CPU: the arithmetic of the order-independent batch sums (csrc/rtk_common.h rtk_stat_add / rtk_stat_read2, tag 1), restated in numpy
float64 -- the device code uses nothing but float64 multiplications by powers of two, floor, differences, additions and fma, all of
which numpy evaluates identically.  An addend is cut into three 30-bit limbs (integers, with its sign); limbs are summed separately;
as long as a limb's sum stays below 2^53 every partial sum is an integer that float64 holds exactly, so any order of addition -- any
interleaving of the workgroups' atomics -- gives the same three numbers.  (The GPU side: tests/test_train_gpu.py, bit-identical train
steps and statistics against float64 BatchNorm from 1e-7 to 1e6.)"""
import math

import numpy as np
import pytest

UNIT = {0: 2.0 ** -36, 1: 2.0 ** -66}          # forward sums (sum w z, sum w z^2) / backward sums (sum dy, sum dy xhat)
PRESCALE = {0: 2.0 ** -24, 1: 2.0 ** 6}        # |x| / (unit 2^60)


def limbs_of(x, kind):
    """-> (l0, l1, l2) float64 integers with the sign of x, or None if x is not finite / too large (the flag word)."""
    m = abs(x) * PRESCALE[kind]
    if not m < 2.0 ** 29:
        return None
    l2 = math.floor(m)
    r1 = (m - l2) * 2.0 ** 30
    l1 = math.floor(r1)
    l0 = math.floor((r1 - l1) * 2.0 ** 30)
    s = -1.0 if x < 0 else 1.0
    return (s * l0, s * l1, s * l2)


def value_of(l0, l1, l2, kind):
    return math.fma(math.fma(l2, 2.0 ** 30, l1), 2.0 ** 30, l0) * UNIT[kind] if hasattr(math, "fma") else ((l2 * 2.0 ** 30 + l1) * 2.0 ** 30 + l0) * UNIT[kind]


def accumulate(xs, kind, order):
    acc = np.zeros(3)
    for i in order:
        q = limbs_of(float(xs[i]), kind)
        assert q is not None
        for l in range(3):
            acc[l] += q[l]                      # float64 addition, as the atomics do
    return acc


@pytest.mark.parametrize("kind,scale", [(0, 1e-6), (0, 1.0), (0, 1e6), (0, 1e12), (1, 1e-15), (1, 1e-6), (1, 1.0), (1, 1e5)])
def test_limb_sums_do_not_depend_on_the_order_and_are_the_exact_sum_of_the_truncated_addends(kind, scale):
    rng = np.random.default_rng(5 + kind)
    xs = rng.standard_normal(4000) * scale * np.exp(rng.uniform(-8, 8, 4000))        # sixteen octaves either side
    xs = xs[np.abs(xs) * PRESCALE[kind] < 2.0 ** 29]
    ref = accumulate(xs, kind, range(len(xs)))
    for seed in range(5):
        perm = np.random.default_rng(seed).permutation(len(xs))
        got = accumulate(xs, kind, perm)
        assert np.array_equal(got, ref)                                             # bit-identical limbs in any order
    assert all(abs(v) < 2.0 ** 53 and v == math.floor(v) for v in ref)
    # each addend is represented to within one unit (truncated towards zero); the limbs add up to exactly the sum of those
    exact = math.fsum(math.copysign(math.floor(abs(float(x)) / UNIT[kind]), float(x)) for x in xs)       # in units (integers: fsum is exact)
    assert ref[2] * 2.0 ** 60 + ref[1] * 2.0 ** 30 + ref[0] == exact or abs((ref[2] * 2.0 ** 60 + ref[1] * 2.0 ** 30 + ref[0]) - exact) <= abs(exact) * 2.0 ** -52
    true = math.fsum(float(x) for x in xs)
    assert abs(value_of(*ref, kind) - true) <= len(xs) * UNIT[kind] + abs(true) * 2.0 ** -50


@pytest.mark.parametrize("kind", [0, 1])
def test_limbs_are_thirty_bit_integers_and_rebuild_the_addend(kind):
    rng = np.random.default_rng(9)
    for x in np.concatenate([rng.standard_normal(500) * 10.0 ** rng.uniform(-12, 6, 500), [0.0, -0.0, UNIT[kind], -UNIT[kind] * 0.999]]):
        q = limbs_of(float(x), kind)
        if q is None:
            continue
        assert all(abs(v) < 2.0 ** 30 and v == math.floor(v) for v in q)
        back = (q[2] * 2.0 ** 60 + q[1] * 2.0 ** 30 + q[0]) * UNIT[kind]
        assert abs(back - x) < UNIT[kind] and abs(back) <= abs(x)                    # truncated towards zero by less than one unit


@pytest.mark.parametrize("kind,limit", [(0, 2.0 ** 53), (1, 2.0 ** 23)])
def test_addends_outside_the_window_are_flagged(kind, limit):
    assert limbs_of(limit * 0.999, kind) is not None
    for bad in (limit, -limit * 4, float("inf"), float("-inf"), float("nan")):
        assert limbs_of(bad, kind) is None


def test_capacity_two_to_the_23_addends_per_limb():
    # the largest limb is 2^30 - 1: 2^23 of them stay below 2^53, where float64 still counts in ones
    assert (2.0 ** 30 - 1) * 2.0 ** 23 < 2.0 ** 53
    big = (2.0 ** 30 - 1) * (2.0 ** 23 - 1)
    assert big + (2.0 ** 30 - 1) == (2.0 ** 30 - 1) * 2.0 ** 23
