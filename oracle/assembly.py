"""TEST INFRASTRUCTURE — Python front of assembly_oracle.cpp: config decoding + ctypes.

The config decoding restates the reference's circe decoders:
  M/model/FeatureSchema.scala:41-81 (type dispatch), M/model/FieldName.scala:38-58,
  M/model/ScopeType.scala:99-110, and the per-feature schema decoders in M/feature/*.scala.
Column order = models.<name>.features order (M/FeatureMapping.scala:66-72,89-99).
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

from . import forest as _forest

KINDS = {k: i for i, k in enumerate(
    ["number", "boolean", "word_count", "vector", "string_index", "string_onehot", "interaction_count", "window_count",
     "rate", "interacted_with", "diversity", "item_age", "position", "relevancy", "const", "item_external", "biencoder",
     "local_time"])}
SCOPES = {"global": 0, "item": 1, "user": 2, "session": 3, "ranking": 4, "item_field": 5, "ranking_field": 6}
MAPPERS = {"time_of_day": 0, "day_of_week": 1, "month_of_year": 2, "year": 3, "second": 4}
NORMS = {"noop": 0, "linear": 1, "position": 2}
INT_MAX = 2147483647


class _ChampNode:
    """one node of scala.collection.immutable.HashMap's prefix tree (Scala 2.13 immutable/HashMap.scala
    BitmapIndexedMapNode): entries and sub-nodes by 5-bit index of the improved hash at this depth"""

    def __init__(self):
        self.entries = {}   # index -> (key, hash)
        self.children = {}  # index -> _ChampNode | list (collision node: equal hashes, insertion order)


def _scala_improve(h: int) -> int:   # collection/Hashing.scala `improve`, Int arithmetic
    m = 0xFFFFFFFF
    h = (h + (~(h << 9) & m)) & m
    h ^= h >> 14
    h = (h + (h << 4)) & m
    return h ^ (h >> 10)


def _java_hash(s: str) -> int:   # String.hashCode: UTF-16 code units
    h = 0
    b = s.encode("utf-16-be")
    for i in range(0, len(b), 2):
        h = (31 * h + ((b[i] << 8) | b[i + 1])) & 0xFFFFFFFF
    return h


def scala_map_key_order(keys):
    """Iteration order of `keys.map(k -> ...).toMap` (InteractedWithFeature.scala:56-65): insertion order up to 4 entries
    (Map1..Map4), a HashMap beyond - built here by inserting the keys one by one the way BitmapIndexedMapNode.updated
    does, then walked like ChampBaseIterator: a node's entries by index, then its sub-nodes by index."""
    if len(keys) <= 4:
        return list(keys)
    root = _ChampNode()

    def insert(node, key, h, shift):
        idx = (h >> shift) & 31
        if idx in node.children:
            child = node.children[idx]
            if isinstance(child, list):
                child.append(key)
            else:
                insert(child, key, h, shift + 5)
        elif idx in node.entries:
            other, oh = node.entries.pop(idx)
            node.children[idx] = merge(other, oh, key, h, shift + 5)
        else:
            node.entries[idx] = (key, h)

    def merge(k0, h0, k1, h1, shift):
        if shift >= 32:
            return [k0, k1]
        n = _ChampNode()
        i0, i1 = (h0 >> shift) & 31, (h1 >> shift) & 31
        if i0 != i1:
            n.entries[i0] = (k0, h0)
            n.entries[i1] = (k1, h1)
        else:
            n.children[i0] = merge(k0, h0, k1, h1, shift + 5)
        return n

    for k in keys:
        insert(root, k, _scala_improve(_java_hash(k)), 0)
    out = []

    def walk(node):
        if isinstance(node, list):
            out.extend(node)
            return
        for idx in sorted(node.entries):
            out.append(node.entries[idx][0])
        for idx in sorted(node.children):
            walk(node.children[idx])

    walk(root)
    return out


def parse_field_name(s: str):
    m = re.fullmatch(r"interaction:([a-zA-Z0-9_]+)\.([a-zA-Z0-9_]+)", s)
    if m:
        return ("interaction:" + m.group(1), m.group(2))
    m = re.fullmatch(r"([a-z\*]+)\.([a-zA-Z0-9_]+)", s)
    if not m:
        raise ValueError(f"cannot decode source field '{s}'")
    src = {"metadata": "item", "item": "item", "user": "user", "ranking": "ranking", "*": "*"}.get(m.group(1))
    if src is None:
        raise ValueError(f"cannot decode source field {m.group(1)}")
    return (src, m.group(2))


def parse_scope(s: str):
    if s in ("global", "item", "user", "session", "ranking"):
        return (SCOPES[s], "")
    m = re.fullmatch(r"item\.([a-zA-Z0-9\-_]+)", s)
    if m:
        return (SCOPES["item_field"], m.group(1))
    m = re.fullmatch(r"ranking\.([a-zA-Z0-9\-_]+)", s)
    if m:
        return (SCOPES["ranking_field"], m.group(1))
    raise ValueError(f"scope type {s} not supported")


_REDUCER_DIM = {"first": 1, "last": 1, "min": 1, "max": 1, "avg": 1, "random": 1, "sum": 1, "size": 1, "euclidean_distance": 1}


def vector_dim(reduce):
    reduce = reduce or ["min", "max", "size", "avg"]
    d = 0
    for r in reduce:
        m = re.fullmatch(r"vector([0-9]+)", r)
        d += int(m.group(1)) if m else 1
    return d


def feature_desc(fc: dict) -> dict:
    """one Metarank feature schema (dict from YAML/JSON) -> oracle feature description"""
    t = fc["type"]
    name = fc["name"]
    d = dict(kind=None, name=name, scope=SCOPES["item"], scope_field="", field="", field_is_ranking=0, dim=1, strs=[],
             top="", bottom="", normalize=0, weight=0.0, div_top=INT_MAX, position=0.0, norm=0, ext_field="", mapper=0)

    def src(*keys):
        for k in keys:
            if k in fc:
                ev, f = parse_field_name(fc[k])
                d["field"] = f
                d["field_is_ranking"] = 1 if ev == "ranking" else 0
                return
        raise ValueError(f"feature {name}: no source field")

    def scope(default=None):
        if "scope" in fc:
            d["scope"], d["scope_field"] = parse_scope(fc["scope"])
        elif default is not None:
            d["scope"], d["scope_field"] = parse_scope(default)
        else:
            raise ValueError(f"feature {name}: scope is required")

    if t == "number":
        d["kind"] = "number"; src("source", "field"); scope()
    elif t == "boolean":
        d["kind"] = "boolean"; src("field", "source"); scope()
    elif t == "word_count":
        d["kind"] = "word_count"; src("source"); scope()
    elif t == "vector":
        d["kind"] = "vector"; src("source"); scope(); d["dim"] = vector_dim(fc.get("reduce"))
    elif t == "string":
        src("source", "field"); scope()
        d["strs"] = list(fc["values"])
        if fc.get("encode") == "index":
            d["kind"] = "string_index"
        else:
            d["kind"] = "string_onehot"; d["dim"] = len(d["strs"])
    elif t == "interaction_count":
        d["kind"] = "interaction_count"; scope()
    elif t == "window_count":
        d["kind"] = "window_count"; scope(); d["dim"] = len(fc["periods"])
    elif t == "rate":
        d["kind"] = "rate"; scope("item"); d["dim"] = len(fc["periods"]); d["top"] = fc["top"]; d["bottom"] = fc["bottom"]
        if d["scope"] not in (SCOPES["item"], SCOPES["item_field"], SCOPES["ranking_field"]):
            raise ValueError(f"scope {fc['scope']} is not supported for rate feature {name}")
        if fc.get("normalize"):
            d["normalize"] = 1; d["weight"] = float(fc["normalize"]["weight"])
    elif t == "interacted_with":
        d["kind"] = "interacted_with"; scope()
        fields = fc["field"] if isinstance(fc["field"], list) else [fc["field"]]
        d["strs"] = scala_map_key_order([parse_field_name(f)[1] for f in fields])
        if fc.get("field_order"):   # a host that sees another order says so
            assert sorted(fc["field_order"]) == sorted(d["strs"])
            d["strs"] = list(fc["field_order"])
        d["dim"] = len(d["strs"])
    elif t == "diversity":
        d["kind"] = "diversity"; src("source"); d["div_top"] = int(fc.get("top", 20))
    elif t == "item_age":
        d["kind"] = "item_age"; src("source")
    elif t == "position":
        d["kind"] = "position"; d["position"] = float(int(fc["position"]))
    elif t == "relevancy":
        d["kind"] = "relevancy"
    elif t == "local_time":
        d["kind"] = "local_time"; src("source"); d["mapper"] = MAPPERS[fc["parse"]]
        if not d["field_is_ranking"]:
            raise ValueError("can only work with ranking event fields")
    elif t == "field_match" and fc.get("method", {}).get("type") == "bi-encoder":
        d["kind"] = "biencoder"; d["norm"] = NORMS[fc.get("norm", "noop")]; d["ext_field"] = "__embedding:" + name
        d["dim"] = 1
    elif t in ("ua", "referer"):
        d["kind"] = "const"; d["dim"] = int(fc["dim"]); d["ext_field"] = "__ext:" + name
    elif t in ("field_match", "random"):
        d["kind"] = "item_external"; d["dim"] = int(fc.get("dim", 1)); d["ext_field"] = "__ext:" + name
    else:
        raise ValueError(f"feature type {t} is not supported")
    return d


def lib():
    L = _forest.lib()
    if not getattr(L, "_asm_bound", False):
        vp, cp, i32, f64 = C.c_void_p, C.c_char_p, C.c_int, C.c_double
        L.orc_store_new.restype = vp
        L.orc_store_free.argtypes = [vp]
        L.orc_store_size.restype = C.c_int64
        L.orc_store_size.argtypes = [vp]
        L.orc_store_delete.argtypes = [vp, cp]
        L.orc_put_double.argtypes = [vp, cp, f64]
        L.orc_put_bool.argtypes = [vp, cp, i32]
        L.orc_put_string.argtypes = [vp, cp, cp]
        L.orc_put_string_list.argtypes = [vp, cp, C.POINTER(cp), i32]
        L.orc_put_double_list.argtypes = [vp, cp, vp, i32]
        L.orc_put_counter.argtypes = [vp, cp, C.c_int64]
        L.orc_put_periodic.argtypes = [vp, cp, vp, i32]
        L.orc_put_bounded_list.argtypes = [vp, cp, C.POINTER(cp), i32]
        L.orc_plan_new.restype = vp
        L.orc_plan_free.argtypes = [vp]
        L.orc_plan_dim.argtypes = [vp]
        L.orc_plan_add.argtypes = [vp, i32, cp, i32, cp, cp, i32, i32, C.POINTER(cp), i32, cp, cp, i32, f64, i32, f64, i32, cp, i32]
        L.orc_assemble.argtypes = [vp, vp, vp, vp]
        L.orc_sort_order.argtypes = [vp, i32, vp]
        L.orc_token_count.argtypes = [cp]
        L.orc_percentile50.restype = f64
        L.orc_percentile50.argtypes = [vp, i32]
        L.orc_java_round.restype = C.c_int64
        L.orc_java_round.argtypes = [f64]
        L.orc_normalize.argtypes = [i32, vp, i32]
        L.orc_cosine.restype = f64
        L.orc_cosine.argtypes = [vp, i32, vp]
        L._asm_bound = True
    return L


def _strs(vals):
    arr = (C.c_char_p * max(len(vals), 1))(*[v.encode() for v in vals])
    return arr


class OracleStore:
    """Map[Key, FeatureValue] (MemKVStore).  Keys are Key.encode strings."""

    def __init__(self):
        self.h = lib().orc_store_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_store_free(self.h)
            self.h = None

    def __len__(self):
        return lib().orc_store_size(self.h)

    def put_double(self, key, v): lib().orc_put_double(self.h, key.encode(), float(v))
    def put_bool(self, key, v): lib().orc_put_bool(self.h, key.encode(), 1 if v else 0)
    def put_string(self, key, v): lib().orc_put_string(self.h, key.encode(), v.encode())
    def put_string_list(self, key, v): lib().orc_put_string_list(self.h, key.encode(), _strs(v), len(v))

    def put_double_list(self, key, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        lib().orc_put_double_list(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), len(a))

    def put_counter(self, key, v): lib().orc_put_counter(self.h, key.encode(), int(v))

    def put_periodic(self, key, v):
        a = np.ascontiguousarray(v, dtype=np.int64)
        lib().orc_put_periodic(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), len(a))

    def put_bounded_list(self, key, v): lib().orc_put_bounded_list(self.h, key.encode(), _strs(v), len(v))
    def delete(self, key): lib().orc_store_delete(self.h, key.encode())


class ReferenceThrows(Exception):
    """the reference's /rank would answer HTTP 500 for this request"""
    CODES = {1: "ArithmeticException: / by zero", 2: "IllegalStateException: dim mismatch", 3: "IllegalArgumentException"}

    def __init__(self, code):
        super().__init__(self.CODES.get(code, str(code)))
        self.code = code


class OraclePlan:
    """FeatureMapping + DatasetDescriptor for one lambdamart model."""

    def __init__(self, config: dict, model: str):
        feats = {f["name"]: f for f in config["features"]}
        self.feature_names = list(config["models"][model]["features"])
        self.h = lib().orc_plan_new()
        self.descs = []
        self.offsets = {}
        for fname in self.feature_names:
            if fname not in feats:
                continue  # FeatureMapping.scala:66-71: unknown names are silently dropped
            d = feature_desc(feats[fname])
            off = lib().orc_plan_add(self.h, KINDS[d["kind"]], d["name"].encode(), d["scope"], d["scope_field"].encode(),
                                     d["field"].encode(), d["field_is_ranking"], d["dim"], _strs(d["strs"]), len(d["strs"]),
                                     d["top"].encode(), d["bottom"].encode(), d["normalize"], d["weight"], d["div_top"],
                                     d["position"], d["norm"], d["ext_field"].encode(), d["mapper"])
            self.offsets[fname] = (off, d["dim"])
            self.descs.append(d)
        self.dim = lib().orc_plan_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_plan_free(self.h)
            self.h = None

    def assemble(self, store: OracleStore, request) -> np.ndarray:
        """request: a built metarank_amd.request.Request (ctypes mrk_request, same layout as the oracle's)."""
        n = request.n_items
        out = np.zeros((n, self.dim), dtype=np.float64)
        st = lib().orc_assemble(self.h, store.h, C.cast(C.byref(request.c), C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if st != 0:
            raise ReferenceThrows(st)
        return out


def sort_order(scores: np.ndarray) -> np.ndarray:
    """Ranker.rerank: sortBy(-score), stable, java.lang.Double.compare."""
    s = np.ascontiguousarray(scores, dtype=np.float64)
    out = np.zeros(len(s), dtype=np.int32)
    lib().orc_sort_order(s.ctypes.data_as(C.c_void_p), len(s), out.ctypes.data_as(C.c_void_p))
    return out


def rerank(plan: OraclePlan, store: OracleStore, forest, request):
    """Ranker.rerank: makeQuery -> predict -> zip/sort.  Returns (matrix, scores, order)."""
    m = plan.assemble(store, request)
    scores = forest.predict(m) if request.n_items else np.zeros(0)
    return m, scores, sort_order(scores)
