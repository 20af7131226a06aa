"""TEST INFRASTRUCTURE — ctypes front of forest_oracle.cpp (see that file's header)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build
from . import formats

_lib = None
_libc = C.CDLL(None)
_libc.strtof.restype = C.c_float
_libc.strtof.argtypes = [C.c_char_p, C.c_void_p]


def strtof(x) -> float:
    """decimal text -> float32 in one rounding (what XGBoost's JSON reader does)."""
    text = getattr(x, "text", None)
    if text is None:
        return float(np.float32(x))
    return float(_libc.strtof(text.encode(), None))


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build())
        L.orc_forest_new.restype = C.c_void_p
        L.orc_forest_new.argtypes = [C.c_int, C.c_float]
        L.orc_forest_free.argtypes = [C.c_void_p]
        L.orc_forest_num_trees.argtypes = [C.c_void_p]
        L.orc_forest_add_lgbm_tree.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_forest_add_xgb_tree.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.orc_forest_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_forest_leaves.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class InfInData(ValueError):
    """XGBoost: 'Input data contains `inf` or a value too large, while `missing` is not set to `inf`'."""


class OracleForest:
    def __init__(self, backend: int, base_score: float = 0.0):
        self.backend = backend
        self.h = lib().orc_forest_new(backend, base_score)
        self.n_features = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_forest_free(self.h)
            self.h = None

    @property
    def n_trees(self) -> int:
        return lib().orc_forest_num_trees(self.h)

    @classmethod
    def from_lightgbm_text(cls, text) -> "OracleForest":
        if isinstance(text, (bytes, bytearray)):
            text = bytes(text).decode("utf-8")
        m = formats.parse_lightgbm_text(text)
        f = cls(0)
        f.n_features = m["max_feature_idx"] + 1
        for t in m["trees"]:
            nl = t["num_leaves"]
            i32 = lambda k: np.ascontiguousarray(t.get(k, []), dtype=np.int32)
            f64 = lambda k: np.ascontiguousarray(t.get(k, []), dtype=np.float64)
            sf, th, dt, lc, rc, lv = i32("split_feature"), f64("threshold"), i32("decision_type"), i32("left_child"), i32("right_child"), f64("leaf_value")
            cb = i32("cat_boundaries")
            ct = np.ascontiguousarray(t.get("cat_threshold", []), dtype=np.uint32)
            lib().orc_forest_add_lgbm_tree(f.h, nl, _p(sf), _p(th), _p(dt), _p(lc), _p(rc), _p(lv), len(cb), _p(cb), len(ct), _p(ct))
        return f

    @classmethod
    def from_xgboost(cls, blob: bytes) -> "OracleForest":
        m = formats.parse_xgboost(bytes(blob))
        f = cls(1, strtof(m["base_score"]))
        f.n_features = m["num_feature"]
        for t in m["trees"]:
            n = len(t["left"])
            i32 = lambda k: np.ascontiguousarray(t[k], dtype=np.int32)
            cond = np.ascontiguousarray([strtof(x) for x in t["split_cond"]], dtype=np.float32)
            offs = np.zeros(n + 1, dtype=np.int32)
            flat = []
            for i, c in enumerate(t["categories"]):
                flat.extend(c)
                offs[i + 1] = len(flat)
            flat = np.ascontiguousarray(flat if flat else [0], dtype=np.int32)
            l, r, si, dl, st = i32("left"), i32("right"), i32("split_index"), i32("default_left"), i32("split_type")
            lib().orc_forest_add_xgb_tree(f.h, n, _p(l), _p(r), _p(si), _p(cond), _p(dl), _p(st), _p(offs), _p(flat))
        return f

    @classmethod
    def from_container(cls, blob: bytes) -> "OracleForest":
        c = formats.parse_container(bytes(blob))
        f = cls.from_lightgbm_text(c["inner"]) if c["booster_tag"] == 0 else cls.from_xgboost(c["inner"])
        f.container_features = c["features"]
        return f

    def predict(self, X: np.ndarray) -> np.ndarray:
        """Booster.predictMat(values, rows, cols)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        rows, cols = X.shape
        out = np.zeros(rows, dtype=np.float64)
        st = lib().orc_forest_predict(self.h, _p(X), rows, cols, _p(out))
        if st == 1:
            raise InfInData("Input data contains `inf` or a value too large, while `missing` is not set to `inf`")
        return out

    def leaves(self, row: np.ndarray) -> np.ndarray:
        row = np.ascontiguousarray(row, dtype=np.float64)
        out = np.zeros(self.n_trees, dtype=np.float64)
        lib().orc_forest_leaves(self.h, _p(row), len(row), _p(out))
        return out
