"""CPU: helper semantics of the assembly oracle pinned on the reference's tests / JVM rules."""
import ctypes as C
import math

import numpy as np

from oracle import assembly as A

NAN = float("nan")


def norm(kind, vals):
    a = np.array(vals, dtype=np.float64)
    A.lib().orc_normalize(kind, a.ctypes.data_as(C.c_void_p), len(a))
    return a


def test_minmax_normalize():  # T/ml/onnx/NormalizeTest.scala:11-31
    assert norm(1, [1.0, 2.0, 3.0]).tolist() == [0.0, 0.5, 1.0]
    r = norm(1, [1.0, 2.0, NAN])
    assert r[:2].tolist() == [0.0, 1.0] and math.isnan(r[2])
    assert all(math.isnan(x) for x in norm(1, [2.0, 2.0]))  # max == min => 0/0


def test_position_normalize():  # T/ml/onnx/NormalizeTest.scala:33-58
    assert norm(2, [1.0, 4.0, 3.0, 2.0, 5.0]).tolist() == [0.0, 0.6, 0.4, 0.2, 0.8]
    r = norm(2, [NAN, 1.0, 4.0, 3.0, 2.0])
    assert math.isnan(r[0]) and r[1:].tolist() == [0.0, 0.6, 0.4, 0.2]


def test_sort_order_is_stable_descending_with_java_double_compare():
    # Ranker.scala:52-67 sortBy(-_.score): NaN last, +0.0 before -0.0 (since -(+0.0) = -0.0 < +0.0), ties keep request order
    s = np.array([1.0, NAN, 3.0, -0.0, 0.0, 3.0, -math.inf, math.inf, NAN])
    assert A.sort_order(s).tolist() == [7, 2, 5, 0, 4, 3, 6, 1, 8]
    assert A.sort_order(np.zeros(5)).tolist() == [0, 1, 2, 3, 4]
    assert A.sort_order(np.zeros(0)).tolist() == []


def test_java_round():
    L = A.lib()
    assert L.orc_java_round(0.5) == 1 and L.orc_java_round(-0.5) == 0 and L.orc_java_round(1.5) == 2
    assert L.orc_java_round(-1.5) == -1 and L.orc_java_round(0.49999999999999994) == 0
    assert L.orc_java_round(NAN) == 0 and L.orc_java_round(1e300) == 2**63 - 1 and L.orc_java_round(-1e300) == -(2**63)
    assert L.orc_java_round(4503599627370497.0) == 4503599627370497


def test_cosine_mixed_precision():  # M/ml/onnx/distance/DistanceFunction.scala:14-26
    q = np.array([0.1, 0.7, -0.3], dtype=np.float32)
    it = np.array([0.2, 0.5, 0.9], dtype=np.float64)
    top = a = b = 0.0
    for i in range(3):
        top += float(q[i]) * it[i]
        a += float(np.float32(q[i] * q[i]))
        b += it[i] * it[i]
    exp = top / (math.sqrt(a) * math.sqrt(b))
    got = A.lib().orc_cosine(q.ctypes.data_as(C.c_void_p), 3, it.ctypes.data_as(C.c_void_p))
    assert got == exp
    z = np.zeros(3)
    assert math.isnan(A.lib().orc_cosine(q.ctypes.data_as(C.c_void_p), 3, z.ctypes.data_as(C.c_void_p)))  # no epsilon


def test_percentile_legacy():
    p = lambda v: A.lib().orc_percentile50(np.array(v, dtype=np.float64).ctypes.data_as(C.c_void_p), len(v))
    assert p([10, 20, 40, 15, 5]) == 15.0 and p([10, 20, 30]) == 20.0 and p([7.0]) == 7.0
    assert p([1, 2, 4, 8]) == 3.0 and p([1, 2]) == 1.5 and p([5, NAN, 1]) == 3.0
    assert math.isnan(p([NAN]))


def test_percentile_legacy_is_the_r6_estimator():
    """commons-math `Percentile.evaluate(50)` (LEGACY) is Hyndman-Fan type 6, pos = p (n + 1) - numpy's method 'weibull', an
    implementation that shares nothing with the oracle: identical for odd n (the median is an element), within an ulp of the neighbours for
    even n (numpy interpolates from the upper neighbour when the weight is >= 0.5, commons-math always from the lower:
    `lower + d * (upper - lower)`, which is what the oracle and the device compute)."""
    rng = np.random.default_rng(5)
    for t in range(1500):
        n = int(rng.integers(1, 80))
        x = rng.normal(size=n) if t % 3 == 0 else rng.integers(0, 10, size=n).astype(float) if t % 3 == 1 else np.round(rng.normal(size=n) * 1e6) / 1e3
        got = A.lib().orc_percentile50(np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), n)
        want = float(np.percentile(x, 50, method="weibull"))
        if n % 2 == 1:
            assert got == want, (n, got, want)
        else:
            lo, hi = np.sort(x)[n // 2 - 1], np.sort(x)[n // 2]
            assert got == lo + 0.5 * (hi - lo) and abs(got - want) <= 2 * np.spacing(max(abs(lo), abs(hi))), (n, got, want)
