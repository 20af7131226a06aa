"""Host-side mirror of ml/onnx/sbert/{OnnxSession,OnnxBiEncoder,OnnxCrossEncoder}.scala for the device path.

    tok = HipTokenizer(open("tokenizer.json", "rb").read())          # HuggingFaceTokenizer.newInstance
    ids, type_ids, mask = tok.encode_batch(["star wars"])             # tokenizer.batchEncode
    enc = HipEncoder(open("pytorch_model.onnx", "rb").read(), tokenizer_json, ctx=ctx)   # OnnxSession.load
    enc.embed(["star wars"])                                          # OnnxBiEncoder.embed  -> f32 [n, dim]
    enc.score_pairs(["query"], ["item title"])                        # OnnxCrossEncoder.encode -> f32 [n]

Everything is computed by libmrk_hip.so (tokenizer on the host, the graph on the GPU); these classes marshal arguments.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .booster import Context, default_context


def _strs(vals):
    bs = [v.encode("utf-8", "surrogatepass") if isinstance(v, str) else v for v in vals]
    return (C.c_char_p * max(len(bs), 1))(*bs), bs


def describe_checkpoint(weights: bytes) -> dict:
    """mrk_checkpoint_describe: names / shapes / sums of the tensors the loader recovers from a model file (host only)."""
    import json
    need = C.c_size_t(0)
    buf = C.create_string_buffer(1 << 20)
    rc = N.lib().mrk_checkpoint_describe(weights, len(weights), buf, len(buf), C.byref(need))
    if rc == N.ERR_INVALID_ARG and need.value > len(buf):
        buf = C.create_string_buffer(need.value)
        rc = N.lib().mrk_checkpoint_describe(weights, len(weights), buf, len(buf), C.byref(need))
    N.check(rc)
    return json.loads(buf.value.decode())


class HipTokenizer:
    def __init__(self, tokenizer_json: bytes | str):
        blob = tokenizer_json.encode() if isinstance(tokenizer_json, str) else tokenizer_json
        self._h = C.c_void_p()
        N.check(N.lib().mrk_tokenizer_load(blob, len(blob), C.byref(self._h)))

    def encode_batch(self, a, b=None, capacity: int = 512):
        n = len(a)
        pa, _ka = _strs(a)
        pb, _kb = _strs(b) if b is not None else (None, None)
        ids = np.zeros((n, capacity), dtype=np.int32)
        types = np.zeros_like(ids)
        mask = np.zeros_like(ids)
        ln = C.c_int(0)
        N.check(N.lib().mrk_tokenizer_encode_batch(self._h, pa, pb, n, ids.ctypes.data, types.ctypes.data, mask.ctypes.data,
                                                  capacity, C.byref(ln)))
        L = ln.value
        flat = lambda x: x.reshape(-1)[: n * L].reshape(n, L).copy()
        return flat(ids), flat(types), flat(mask)

    def close(self):
        if self._h:
            N.lib().mrk_tokenizer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipEncoder:
    def __init__(self, weights: bytes, tokenizer_json: bytes | str, heads: int = 0, ctx: Context | None = None, f32: bool | None = None,
                 precision: str | None = None):
        """precision (mrk_encoder_load_ex): "f32" (default, what mrk_encoder_load uses: f32 operands on the f32-input matrix
        instruction - the fp32 ONNX session's arithmetic, batch-independent bits; `f32=True` is the same), "f16" (opt-in:
        fp16 operands on the matrix cores, ~3.5x faster per packed batch, cosines within 3e-3), "auto" (ABI <= 7 name: f32 now)"""
        self.ctx = ctx or default_context()
        tj = tokenizer_json.encode() if isinstance(tokenizer_json, str) else tokenizer_json
        self._h = C.c_void_p()
        self.precision = precision or ("f16" if f32 is False else "f32")
        code = {"f16": 0, "f32": 1, "auto": 2}[self.precision]
        N.check(N.lib().mrk_encoder_load_ex(self.ctx.handle, weights, len(weights), tj, len(tj), heads, code, C.byref(self._h)))
        info = N.mrk_encoder_info()
        N.check(N.lib().mrk_encoder_get_info(self._h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in info._fields_}
        self.dim = info.hidden

    @property
    def handle(self):
        return self._h

    def _ids(self, ids, type_ids, mask):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        type_ids = np.zeros_like(ids) if type_ids is None else np.ascontiguousarray(type_ids, dtype=np.int32)
        return ids, type_ids, mask

    def hidden_ids(self, ids, type_ids, mask) -> np.ndarray:
        ids, type_ids, mask = self._ids(ids, type_ids, mask)
        out = np.empty(ids.shape + (self.dim,), dtype=np.float32)
        N.check(N.lib().mrk_encoder_hidden_ids(self._h, ids.ctypes.data, type_ids.ctypes.data, mask.ctypes.data, ids.shape[0], ids.shape[1], out.ctypes.data))
        return out

    def embed_ids(self, ids, type_ids, mask) -> np.ndarray:
        ids, type_ids, mask = self._ids(ids, type_ids, mask)
        out = np.empty((ids.shape[0], self.dim), dtype=np.float32)
        N.check(N.lib().mrk_encoder_embed_ids(self._h, ids.ctypes.data, type_ids.ctypes.data, mask.ctypes.data, ids.shape[0], ids.shape[1], out.ctypes.data))
        return out

    def score_ids(self, ids, type_ids, mask) -> np.ndarray:
        ids, type_ids, mask = self._ids(ids, type_ids, mask)
        out = np.empty(ids.shape[0], dtype=np.float32)
        N.check(N.lib().mrk_encoder_score_ids(self._h, ids.ctypes.data, type_ids.ctypes.data, mask.ctypes.data, ids.shape[0], ids.shape[1], out.ctypes.data))
        return out

    def embed(self, texts) -> np.ndarray:
        p, _k = _strs(texts)
        out = np.empty((len(texts), self.dim), dtype=np.float32)
        N.check(N.lib().mrk_encoder_embed(self._h, p, len(texts), out.ctypes.data))
        return out

    def score_pairs(self, a, b) -> np.ndarray:
        pa, _ka = _strs(a)
        pb, _kb = _strs(b)
        out = np.empty(len(a), dtype=np.float32)
        N.check(N.lib().mrk_encoder_score_pairs(self._h, pa, pb, len(a), out.ctypes.data))
        return out

    def close(self):
        if self._h:
            N.lib().mrk_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
