"""Host-side mirror of ltrlib's Booster[_] for the device scorer.

Reference interface (io.github.metarank.ltrlib Booster, used at
ml/rank/LambdaMARTRanker.scala:348,362,365,373 and constructed at :229-230):
    predictMat(values: Array[Double], rows: Int, cols: Int): Array[Double]
    save(): Array[Byte]      weights(): Array[Double] (:392)      close(): Unit      isClosed(): Boolean
`HipBooster(bytes, backend)` is the drop-in for LightGBMBooster(bytes) / XGBoostBooster(bytes);
all arithmetic happens in libmrk_hip.so (hand-written gfx950 kernels) — there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

LIGHTGBM, XGBOOST = 0, 1


def inspect_weights(model_bytes: bytes, backend: int, n_cols: int, importance_type: int = 1) -> np.ndarray:
    """mrk_model_inspect_weights: Booster.weights() from booster bytes on the host (no context, no device)"""
    out = np.zeros(max(int(n_cols), 1), dtype=np.float64)
    N.check(N.lib().mrk_model_inspect_weights(backend, model_bytes, len(model_bytes), importance_type, out.ctypes.data_as(C.c_void_p), int(n_cols)))
    return out[:int(n_cols)]


class Context:
    """mrk_ctx: one per device; several may live in one process (create_many), each driven by its own host thread."""

    def __init__(self, device: int = 0, _handle=None):
        self.device = device
        if _handle is not None:
            self._h = _handle
            return
        self._h = C.c_void_p()
        ids = (C.c_int * 1)(device)
        N.check(N.lib().mrk_init(ids, 1, C.byref(self._h)))

    @staticmethod
    def device_count() -> int:
        return N.lib().mrk_device_count()

    @classmethod
    def create_many(cls, devices) -> list["Context"]:
        """mrk_init(device_ids, n): one context per listed ordinal, all in THIS process (HipConfig(devices: List[Int]))"""
        n = len(devices)
        ids = (C.c_int * n)(*devices)
        hs = (C.c_void_p * n)()
        N.check(N.lib().mrk_init(ids, n, hs))
        return [cls(d, _handle=C.c_void_p(h)) for d, h in zip(devices, hs)]

    @staticmethod
    def comm_init_local(ctxs) -> None:
        """mrk_comm_init_local: the contexts of this process become ranks 0..n-1 of one RCCL communicator"""
        hs = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        N.check(N.lib().mrk_comm_init_local(hs, len(ctxs)))

    @property
    def handle(self):
        if not self._h:
            raise N.MrkError(N.ERR_INVALID_ARG, "context is closed")
        return self._h

    def sync(self):
        N.check(N.lib().mrk_sync(self.handle))

    def profile_enable(self, on: bool = True):
        N.check(N.lib().mrk_profile_enable(self.handle, 1 if on else 0))

    def profile_get(self, kernel: str):
        ms, n = C.c_double(), C.c_int64()
        N.check(N.lib().mrk_profile_get(self.handle, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- multi-GPU: the RCCL communicator lives inside the library (csrc/comm.cpp)
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: rank 0 draws it and hands the 128 bytes to the other ranks"""
        buf = (C.c_uint8 * 128)()
        N.check(N.lib().mrk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid: bytes, rank: int, world: int):
        """ncclCommInitRank on this context's device: collective over all `world` ranks"""
        N.check(N.lib().mrk_comm_init(self.handle, uid, rank, world))

    @property
    def comm_rank(self) -> int:
        return N.lib().mrk_comm_rank(self.handle)

    @property
    def comm_world(self) -> int:
        return N.lib().mrk_comm_world(self.handle)

    def comm_max(self, v: float) -> float:
        x = C.c_double(v)
        N.check(N.lib().mrk_comm_max_f64(self.handle, C.byref(x)))
        return x.value

    def comm_barrier(self):
        N.check(N.lib().mrk_comm_barrier(self.handle))

    def close(self):
        if self._h:
            N.lib().mrk_shutdown(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class HipBooster:
    def __init__(self, model_bytes: bytes, backend: int, ctx: Context | None = None, _handle=None):
        self.ctx = ctx or default_context()
        self._bytes = bytes(model_bytes)
        self.backend = backend
        if _handle is not None:
            self._h = _handle
        else:
            self._h = C.c_void_p()
            buf = (C.c_ubyte * len(self._bytes)).from_buffer_copy(self._bytes) if self._bytes else None
            N.check(N.lib().mrk_model_load(self.ctx.handle, backend, buf, len(self._bytes), C.byref(self._h)))

    @classmethod
    def from_container(cls, blob: bytes, feature_names=None, ctx: Context | None = None) -> "HipBooster":
        """LambdaMARTPredictor.load(bytes) (LambdaMARTRanker.scala:192-236)."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        blob = bytes(blob)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        if feature_names is not None:
            arr = (C.c_char_p * len(feature_names))(*[f.encode() for f in feature_names])
            st = N.lib().mrk_model_load_container(ctx.handle, buf, len(blob), arr, len(feature_names), C.byref(h))
        else:
            st = N.lib().mrk_model_load_container(ctx.handle, buf, len(blob), None, 0, C.byref(h))
        N.check(st)
        b = cls.__new__(cls)
        b.ctx, b._bytes, b._h = ctx, blob, h
        b.backend = b.info()["backend"]
        return b

    @property
    def handle(self):
        if not self._h:
            raise N.MrkError(N.ERR_INVALID_ARG, "booster is closed")
        return self._h

    def predictMat(self, values, rows: int, cols: int) -> np.ndarray:
        x = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        if x.size != rows * cols:
            raise N.MrkError(N.ERR_INVALID_ARG, f"values has {x.size} cells, expected {rows}x{cols}")
        out = np.empty(rows, dtype=np.float64)
        N.check(N.lib().mrk_model_predict_f64(self.handle, x.ctypes.data_as(C.c_void_p), rows, cols,
                                               out.ctypes.data_as(C.c_void_p)))
        return out

    def predict(self, X) -> np.ndarray:
        X = np.ascontiguousarray(X, dtype=np.float64)
        return self.predictMat(X, X.shape[0], X.shape[1])

    def info(self) -> dict:
        inf = N.mrk_model_info()
        N.check(N.lib().mrk_model_get_info(self.handle, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def save(self) -> bytes:
        return self._bytes

    GAIN, SPLIT, TOTAL_GAIN = 1, 0, 2

    def weights(self, n_cols: int | None = None, importance_type: int = 1) -> np.ndarray:
        """Booster.weights() (LambdaMARTRanker.scala:391-392): the library's feature importance per matrix column, mrk_model_weights"""
        n = self.info()["n_features"] if n_cols is None else int(n_cols)
        out = np.zeros(max(n, 1), dtype=np.float64)
        N.check(N.lib().mrk_model_weights(self.handle, importance_type, out.ctypes.data_as(C.c_void_p), n))
        return out[:n]

    def close(self):
        if self._h:
            N.lib().mrk_model_free(self._h)
            self._h = C.c_void_p()

    def isClosed(self) -> bool:
        return not bool(self._h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
