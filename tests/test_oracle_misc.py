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
