"""Host-side mirror of ml/Ranker.scala for the device path.

    ranker = HipRanker(config)                 # FeatureMapping.fromFeatureSchema (config = features + models)
    ranker.put_double("item=42/popularity", 7) # KVStore.put(Key, FeatureValue)   (Key.encode strings)
    m, scores, order = ranker.rerank("xgboost", ranking_event, booster, explain=True)   # Ranker.rerank

Everything is computed by libmrk_hip.so on the GPU; this class only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import _native as N
from .booster import Context, HipBooster, default_context
from .request import Request, RequestSet, request_array


def _strs(vals):
    bs = [v.encode() for v in vals]
    return (C.c_char_p * max(len(bs), 1))(*bs), bs


class Batch:
    """mrk_batch: a prepared, device-resident batch of requests."""

    def __init__(self, ranker: "HipRanker", model_name: str | None = None, events=None):
        """With events: mrk_batch_prepare (resolve + upload once).  Without: mrk_batch_create - an empty batch for the
        serving loop, (re)filled with load()."""
        self.ranker = ranker
        self._h = C.c_void_p()
        self.requests, self.n_req, self.total_items, self.dim, self.offsets = [], 0, 0, 0, np.zeros(1, dtype=np.int64)
        if events is None:
            N.check(N.lib().mrk_batch_create(ranker.ctx.handle, C.byref(self._h)))
            return
        self.requests = [e if isinstance(e, Request) else Request(e) for e in events]
        arr = request_array(self.requests)
        N.check(N.lib().mrk_batch_prepare(ranker.ctx.handle, model_name.encode(), arr, len(self.requests), C.byref(self._h)))
        self.n_req = len(self.requests)
        self.total_items = N.lib().mrk_batch_total_items(self._h)
        self.dim = ranker.dim(model_name)
        self.offsets = np.concatenate([[0], np.cumsum([r.n_items for r in self.requests])]).astype(np.int64)

    def load(self, model_name: str, rs: "RequestSet | list", flat: bool = True):
        """mrk_batch_load: (re)fill the batch.  A RequestSet travels as flat id bytes (device-side id resolution) unless
        flat=False; a list of events / Requests goes through the per-item C strings (host lookups)."""
        if isinstance(rs, RequestSet):
            if not flat:
                raise ValueError("a RequestSet carries no per-item C strings")
            N.check(N.lib().mrk_batch_load(self._h, model_name.encode(), rs.arr, rs.n_req, C.byref(rs.ids)))
            self.requests, self.n_req, self.offsets = rs.requests, rs.n_req, rs.offsets_per_request
            self._keep = rs
        else:
            self.requests = [e if isinstance(e, Request) else Request(e) for e in rs]
            arr = request_array(self.requests)
            N.check(N.lib().mrk_batch_load(self._h, model_name.encode(), arr, len(self.requests), None))
            self.n_req = len(self.requests)
            self.offsets = np.concatenate([[0], np.cumsum([r.n_items for r in self.requests])]).astype(np.int64)
            self._keep = arr
        self.total_items = N.lib().mrk_batch_total_items(self._h)
        self.dim = self.ranker.dim(model_name)

    def enqueue_fetch(self):
        N.check(N.lib().mrk_batch_enqueue_fetch(self._h))

    def host_outputs(self):
        """waits for the batch; (scores, order, status) as numpy VIEWS of the batch's pinned result buffer"""
        s, o, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(N.lib().mrk_batch_host_outputs(self._h, C.byref(s), C.byref(o), C.byref(st)))
        T, R = max(self.total_items, 1), max(self.n_req, 1)
        scores = np.ctypeslib.as_array((C.c_double * T).from_address(s.value))[:self.total_items]
        order = np.ctypeslib.as_array((C.c_int32 * T).from_address(o.value))[:self.total_items]
        status = np.ctypeslib.as_array((C.c_int32 * R).from_address(st.value))[:self.n_req]
        return scores, order, status

    def run(self, booster: HipBooster | None):
        N.check(N.lib().mrk_batch_run(self._h, booster.handle if booster is not None else None))

    def run_shard(self, booster: HipBooster | None, shard_index: int, shard_count: int):
        """assemble + score batch items [index * chunk, (index + 1) * chunk) only; no sort (mrk_batch_run_shard)"""
        N.check(N.lib().mrk_batch_run_shard(self._h, booster.handle if booster is not None else None, shard_index, shard_count))

    def shard_chunk(self, shard_count: int) -> int:
        c = N.lib().mrk_batch_shard_chunk(self._h, shard_count)
        if c < 0:
            N.check(c)
        return c

    def run_sharded(self, booster: HipBooster | None):
        """mrk_batch_run_sharded: this rank's slice -> one in-place RCCL all-gather of the scores -> sort"""
        N.check(N.lib().mrk_batch_run_sharded(self._h, booster.handle if booster is not None else None))

    def allgather_scores(self):
        N.check(N.lib().mrk_batch_allgather_scores(self._h))

    def gather_scores(self) -> int:
        """mrk_batch_gather_scores: device pointer of world x total_items f64 (every rank's scores)"""
        p = C.c_void_p()
        N.check(N.lib().mrk_batch_gather_scores(self._h, C.byref(p)))
        return p.value

    def sort(self):
        N.check(N.lib().mrk_batch_sort(self._h))

    def sync(self):
        N.check(N.lib().mrk_batch_sync(self._h))

    @property
    def stream(self):
        return N.lib().mrk_batch_stream(self._h)

    def device_outputs(self, matrix: bool = False):
        """device pointers (scores, order, matrix | None); asking for the matrix makes later runs write it"""
        s, o, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(N.lib().mrk_batch_device_outputs(self._h, C.byref(s), C.byref(o), C.byref(m) if matrix else None))
        return s.value, o.value, (m.value if matrix else None)

    def fetch(self, matrix: bool = False):
        scores = np.empty(self.total_items, dtype=np.float64)
        order = np.empty(self.total_items, dtype=np.int32)
        mat = np.empty((self.total_items, self.dim), dtype=np.float64) if matrix else None
        N.check(N.lib().mrk_batch_fetch(self._h, scores.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p),
                                        mat.ctypes.data_as(C.c_void_p) if matrix else None))
        return scores, order, mat

    def status(self) -> np.ndarray:
        st = np.zeros(max(self.n_req, 1), dtype=np.int32)
        N.check(N.lib().mrk_batch_status(self._h, st.ctypes.data_as(C.c_void_p)))
        return st[:self.n_req]

    def close(self):
        if self._h:
            N.lib().mrk_batch_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipRanker:
    def __init__(self, config: dict, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self._own_ctx = False
        if getattr(self.ctx, "_configured", False):
            # one configuration per mrk_ctx: give this ranker a private context on the same device
            self.ctx = Context(self.ctx.device)
            self._own_ctx = True
        self.config = config
        blob = json.dumps({"features": config["features"], "models": config.get("models", {})}).encode()
        N.check(N.lib().mrk_config_load_json(self.ctx.handle, blob, len(blob)))
        self.ctx._configured = True

    # ---- KVStore.put
    def _k(self, key): return key.encode()
    def put_double(self, key, v): N.check(N.lib().mrk_store_put_double(self.ctx.handle, self._k(key), float(v)))
    def put_bool(self, key, v): N.check(N.lib().mrk_store_put_bool(self.ctx.handle, self._k(key), 1 if v else 0))
    def put_string(self, key, v): N.check(N.lib().mrk_store_put_string(self.ctx.handle, self._k(key), v.encode()))

    def put_string_list(self, key, v):
        arr, _keep = _strs(v)
        N.check(N.lib().mrk_store_put_string_list(self.ctx.handle, self._k(key), arr, len(v)))

    def put_double_list(self, key, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        N.check(N.lib().mrk_store_put_double_list(self.ctx.handle, self._k(key), a.ctypes.data_as(C.c_void_p), len(a)))

    def put_counter(self, key, v): N.check(N.lib().mrk_store_put_counter(self.ctx.handle, self._k(key), int(v)))

    def put_periodic(self, key, v):
        a = np.ascontiguousarray(v, dtype=np.int64)
        N.check(N.lib().mrk_store_put_periodic(self.ctx.handle, self._k(key), a.ctypes.data_as(C.c_void_p), len(a)))

    def put_bounded_list(self, key, v):
        arr, _keep = _strs(v)
        N.check(N.lib().mrk_store_put_bounded_list(self.ctx.handle, self._k(key), arr, len(v)))

    # ---- write path: raw Writes (FeatureValueFlow.commitWrite); the FeatureValue is derived by the library
    def increment_periodic(self, key, ts_ms, inc=1):
        N.check(N.lib().mrk_store_increment_periodic(self.ctx.handle, self._k(key), int(ts_ms), int(inc)))

    def increment(self, key, inc=1):
        N.check(N.lib().mrk_store_increment(self.ctx.handle, self._k(key), int(inc)))

    def append(self, key, value, ts_ms):
        N.check(N.lib().mrk_store_append(self.ctx.handle, self._k(key), str(value).encode(), int(ts_ms)))

    def put_binary(self, blob: bytes, now_ms: int | None = None) -> int:
        """bulk load: concatenated FeatureValueCodec records (the reference's binary wire format); now_ms: remember every
        record's deadline now_ms + expire (mrk_store_put_binary_at) for a later expire()"""
        n = C.c_int(0)
        if now_ms is None:
            N.check(N.lib().mrk_store_put_binary(self.ctx.handle, blob, len(blob), C.byref(n)))
        else:
            N.check(N.lib().mrk_store_put_binary_at(self.ctx.handle, blob, len(blob), now_ms, C.byref(n)))
        return n.value

    def expire(self, now_ms: int) -> int:
        """mrk_store_expire: drops the values whose deadline (FeatureValue.expire after their last write) has passed"""
        n = C.c_int64(0)
        N.check(N.lib().mrk_store_expire(self.ctx.handle, now_ms, C.byref(n)))
        return n.value

    def clone_items(self, copies: int) -> int:
        """measurement aid (mrk_debug_clone_items): every item gets `copies` deep copies under the ids '<id>#k'"""
        fn = N.lib().mrk_debug_clone_items
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        n = C.c_int64(0)
        N.check(fn(self.ctx.handle, int(copies), C.byref(n)))
        return n.value

    def store_info(self, scope: int = 1) -> dict:
        """layout of one scope's table (1 = items): slots, record stride, inline heap, pool sizes (mrk_debug_store_info)"""
        fn = N.lib().mrk_debug_store_info
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        out = (C.c_int64 * 6)()
        N.check(fn(self.ctx.handle, scope, out))
        return dict(zip(("slots", "stride", "heap_off", "heap_cap", "tok_pool", "f64_pool"), [int(x) for x in out]))

    def item_stride(self) -> int:
        return self.store_info(1)["stride"]

    def delete(self, key): N.check(N.lib().mrk_store_delete(self.ctx.handle, self._k(key)))
    def flush(self): N.check(N.lib().mrk_store_flush(self.ctx.handle))

    def dim(self, model_name: str) -> int:
        d = N.lib().mrk_model_dim(self.ctx.handle, model_name.encode())
        if d < 0:
            N.check(d)
        return d

    def load_model(self, blob: bytes, backend: int) -> HipBooster:
        return HipBooster(blob, backend, self.ctx)

    # ---- Ranker.rerank
    def rerank(self, model_name: str, event, booster: HipBooster | None = None, explain: bool = False):
        """-> (matrix | None, scores, order); order[k] = request index of the k-th response item."""
        req = event if isinstance(event, Request) else Request(event)
        n = req.n_items
        dim = self.dim(model_name)
        scores = np.empty(n, dtype=np.float64)
        order = np.empty(n, dtype=np.int32)
        mat = np.empty((n, dim), dtype=np.float64) if explain else None
        N.check(N.lib().mrk_rank(self.ctx.handle, booster.handle if booster is not None else None, model_name.encode(),
                                 C.byref(req.c), scores.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p),
                                 mat.ctypes.data_as(C.c_void_p) if explain else None))
        return mat, scores, order

    def rerank_binary(self, model_name: str, event_bytes: bytes, booster: HipBooster | None, capacity: int = 4096):
        """Ranker.rerank for a request in the reference's binary RankingEventFormat -> (scores, order)"""
        scores = np.empty(capacity, dtype=np.float64)
        order = np.empty(capacity, dtype=np.int32)
        n = C.c_int(0)
        N.check(N.lib().mrk_rank_binary(self.ctx.handle, booster.handle if booster is not None else None, model_name.encode(),
                                        event_bytes, len(event_bytes), C.byref(n), scores.ctypes.data_as(C.c_void_p),
                                        order.ctypes.data_as(C.c_void_p), capacity))
        return scores[:n.value], order[:n.value]

    def warmup(self, model_name: str, booster: HipBooster) -> int:
        n = C.c_int(0)
        N.check(N.lib().mrk_model_warmup(self.ctx.handle, booster.handle, model_name.encode(), C.byref(n)))
        return n.value

    def bind_encoder(self, feature: str, encoder):
        """mrk_config_bind_encoder: the bi-encoder `feature` embeds its rankingField text on the device from now on"""
        N.check(N.lib().mrk_config_bind_encoder(self.ctx.handle, feature.encode(), encoder.handle))

    def serve(self, model_name: str, booster: HipBooster, n_slots: int = 4) -> "Server":
        """mrk_serve_start: the serving queue (persistent workgroups polling slots in pinned memory)"""
        return Server(self, model_name, booster, n_slots)

    def warmup_kernels(self, model_name: str):
        """mrk_config_warmup: waits for the background compiles of this model's specialised kernels that are under way"""
        N.check(N.lib().mrk_config_warmup(self.ctx.handle, model_name.encode()))

    def kernel_keys(self, model_name: str) -> dict:
        """mrk_config_kernel_keys: {kernel name: "<key> program|program+forest"} of this model's loaded specialised kernels"""
        need = C.c_size_t(0)
        N.lib().mrk_config_kernel_keys(self.ctx.handle, model_name.encode(), None, 0, C.byref(need))
        buf = C.create_string_buffer(max(need.value, 1))
        N.check(N.lib().mrk_config_kernel_keys(self.ctx.handle, model_name.encode(), buf, len(buf), C.byref(need)))
        out = {}
        for ln in buf.value.decode().splitlines():
            name, _, rest = ln.partition(" ")
            out.setdefault(name, []).append(rest)
        return out

    def prepare(self, model_name: str, events) -> Batch:
        return Batch(self, model_name, events)

    def new_batch(self) -> Batch:
        """mrk_batch_create: an empty, reusable batch (own stream) for the serving loop"""
        return Batch(self)

    def close(self):
        if self._own_ctx:
            self.ctx.close()
            self._own_ctx = False


class Server:
    """mrk_serve_*: Ranker.rerank through the serving queue - no launch, no copy command on the request path."""

    def __init__(self, ranker: HipRanker, model_name: str, booster: HipBooster, n_slots: int = 4):
        self.ranker = ranker
        self._h = C.c_void_p()
        N.check(N.lib().mrk_serve_start(ranker.ctx.handle, booster.handle, model_name.encode(), n_slots, C.byref(self._h)))

    def rerank(self, event):
        """-> (scores, order) like HipRanker.rerank"""
        req = event if isinstance(event, Request) else Request(event)
        n = req.n_items
        scores = np.empty(n, dtype=np.float64)
        order = np.empty(n, dtype=np.int32)
        N.check(N.lib().mrk_serve_rank(self._h, C.byref(req.c), scores.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p)))
        return scores, order

    def stats(self) -> dict:
        out = (C.c_int64 * 10)()
        N.check(N.lib().mrk_serve_stats(self._h, out, 10))
        n = max(out[0], 1)
        return {"queue": out[0], "fallback": out[1], "launches": out[2], "device_rank_mhz": out[9] / max(out[7], 1) * 1e3,
                "us_per_request": {"host_resolve_pack": out[3] / n / 1e3, "host_publish_to_ack": out[4] / n / 1e3, "host_copy_out": out[5] / n / 1e3,
                                   "device_input": out[6] / n / 1e3, "device_rank": out[7] / n / 1e3, "device_write_back": out[8] / n / 1e3}}

    def close(self):
        if self._h:
            N.lib().mrk_serve_stop(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
