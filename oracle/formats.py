"""TEST INFRASTRUCTURE — model-format readers of the CPU oracle (pure Python, independent of the
product's C++ readers in metarank_amd/csrc/forest.cpp).

Formats:
  * Metarank model container, bitstream v2/v3 — written by
    /root/reference/src/main/scala/ai/metarank/ml/rank/LambdaMARTRanker.scala:367-389 and read by
    :192-236 (java.io.DataOutputStream => big-endian, writeUTF = u16 length + modified UTF-8).
  * LightGBM model string (inner bytes when boosterTag == 0) — third-party, LightGBM 4.6.0
    (lightgbm4j 4.6.0-1, reference build.sbt:58), GBDT::SaveModelToString / Tree::ToString.
  * XGBoost JSON / UBJSON (inner bytes when boosterTag == 1) — third-party, xgboost4j pulled by
    ltrlib 0.2.6 (reference build.sbt:57), doc/tutorials/saving_model.rst schema.
"""
from __future__ import annotations

import json
import struct
from typing import Any


# --------------------------------------------------------------------------- Metarank container
def parse_container(blob: bytes) -> dict:
    pos = 0
    version = blob[pos]
    pos += 1
    if version not in (2, 3):
        raise ValueError(f"unsupported bitstream version {version}")
    (nfeat,) = struct.unpack_from(">i", blob, pos)
    pos += 4
    features = []
    for _ in range(nfeat):
        (n,) = struct.unpack_from(">H", blob, pos)
        pos += 2
        features.append(blob[pos:pos + n].decode("utf-8"))
        pos += n
    tag = blob[pos]
    pos += 1
    if tag not in (0, 1):
        raise ValueError(f"unsupported booster tag {tag}")
    (size,) = struct.unpack_from(">i", blob, pos)
    pos += 4
    inner = blob[pos:pos + size]
    if len(inner) != size:
        raise ValueError("truncated container")
    return {"version": version, "features": features, "booster_tag": tag, "inner": inner}


# --------------------------------------------------------------------------- LightGBM text
def parse_lightgbm_text(text: str) -> dict:
    header: dict[str, str] = {}
    trees: list[dict] = []
    cur: dict[str, str] | None = None
    for raw in text.split("\n"):
        line = raw.rstrip("\r")
        if line.startswith("Tree="):
            if cur is not None:
                trees.append(cur)
            cur = {}
            continue
        if line == "end of trees":
            break
        if not line:
            continue
        if "=" in line:
            k, v = line.split("=", 1)
            if cur is None:
                header[k] = v
            else:
                cur[k] = v
        elif cur is None:
            header[line] = ""
    if cur is not None:
        trees.append(cur)

    out_trees = []
    for t in trees:
        nl = int(t["num_leaves"])
        tree: dict[str, Any] = {"num_leaves": nl, "leaf_value": [float(x) for x in t["leaf_value"].split()]}
        if nl > 1:
            tree["split_feature"] = [int(x) for x in t["split_feature"].split()]
            tree["threshold"] = [float(x) for x in t["threshold"].split()]
            tree["decision_type"] = [int(x) for x in t["decision_type"].split()]
            tree["left_child"] = [int(x) for x in t["left_child"].split()]
            tree["right_child"] = [int(x) for x in t["right_child"].split()]
        if int(t.get("num_cat", "0")) > 0:
            tree["cat_boundaries"] = [int(x) for x in t["cat_boundaries"].split()]
            tree["cat_threshold"] = [int(x) for x in t["cat_threshold"].split()]
        out_trees.append(tree)
    return {
        "max_feature_idx": int(header.get("max_feature_idx", "-1")),
        "objective": header.get("objective", ""),
        "trees": out_trees,
    }


# --------------------------------------------------------------------------- UBJSON
class _Ubj:
    def __init__(self, b: bytes):
        self.b = b
        self.p = 0

    def u8(self) -> int:
        v = self.b[self.p]
        self.p += 1
        return v

    def num(self, m: int):
        fmt = {ord("i"): ">b", ord("U"): ">B", ord("I"): ">h", ord("l"): ">i", ord("L"): ">q",
               ord("d"): ">f", ord("D"): ">d"}[m]
        (v,) = struct.unpack_from(fmt, self.b, self.p)
        self.p += struct.calcsize(fmt)
        return v

    def string(self) -> str:
        n = self.num(self.u8())
        s = self.b[self.p:self.p + n].decode("utf-8")
        self.p += n
        return s

    def value(self, m: int):
        c = chr(m)
        if c in "ZN":
            return None
        if c == "T":
            return True
        if c == "F":
            return False
        if c in "iUIlLdD":
            return self.num(m)
        if c == "C":
            return chr(self.u8())
        if c in "SH":
            return self.string()
        if c == "[":
            et, cnt = None, None
            if self.b[self.p] == ord("$"):
                self.p += 1
                et = self.u8()
            if self.b[self.p] == ord("#"):
                self.p += 1
                cnt = self.num(self.u8())
            out = []
            if cnt is not None:
                for _ in range(cnt):
                    out.append(self.value(et if et is not None else self.u8()))
            else:
                while True:
                    mm = self.u8()
                    if mm == ord("]"):
                        break
                    out.append(self.value(mm))
            return out
        if c == "{":
            et, cnt = None, None
            if self.b[self.p] == ord("$"):
                self.p += 1
                et = self.u8()
            if self.b[self.p] == ord("#"):
                self.p += 1
                cnt = self.num(self.u8())
            out = {}
            if cnt is not None:
                for _ in range(cnt):
                    k = self.string()
                    out[k] = self.value(et if et is not None else self.u8())
            else:
                while self.b[self.p] != ord("}"):
                    k = self.string()
                    out[k] = self.value(self.u8())
                self.p += 1
            return out
        raise ValueError(f"ubjson: unknown marker {c!r}")


def parse_ubjson(b: bytes):
    u = _Ubj(b)
    return u.value(u.u8())


# --------------------------------------------------------------------------- XGBoost
def _num(x) -> float:
    """a scalar as the XGBoost writers spell it: a number, "5E-1", "[5E-1]" (3.x: one value per target) or a one-element array"""
    if isinstance(x, (list, tuple)):
        if len(x) != 1:
            raise ValueError("multi-target models are not supported")
        return _num(x[0])
    if isinstance(x, str):
        if "," in x:
            raise ValueError("multi-target models are not supported")
        return float(x.strip("[]"))
    return float(x)


def parse_xgboost_legacy(blob: bytes) -> dict:
    """XGBoost's legacy binary serialisation (learner.cc LearnerImpl::LoadModel -> gbtree_model.cc GBTreeModel::Load ->
    tree_model.cc RegTree::Load of XGBoost 1.x): ["binf"], LearnerModelParamLegacy (136 B: f32 base_score, u32
    num_feature, ...), two u64-length-prefixed names (objective, booster), GBTreeModelParam (160 B: i32 num_trees first,
    i32 size_leaf_vector at +28), per tree TreeParam (148 B: i32 num_nodes at +4, i32 size_leaf_vector at +20) +
    num_nodes x Node {i32 parent, i32 cleft, i32 cright, u32 sindex, f32 info} + num_nodes x 16 B stats [+ leaf vector],
    then num_trees x i32 tree_info."""
    import struct

    at = 4 if blob[:4] == b"binf" else 0
    base_score, num_feature = struct.unpack_from("<fI", blob, at)
    at += 136
    names = []
    for _ in range(2):
        (n,) = struct.unpack_from("<Q", blob, at)
        names.append(blob[at + 8:at + 8 + n].decode())
        at += 8 + n
    if names[1] != "gbtree":
        raise ValueError("only gbtree boosters")
    (num_trees,) = struct.unpack_from("<i", blob, at)
    at += 160
    trees = []
    for _ in range(num_trees):
        (num_nodes,) = struct.unpack_from("<i", blob, at + 4)
        (leaf_vec,) = struct.unpack_from("<i", blob, at + 20)
        at += 148
        t = {"left": [], "right": [], "split_index": [], "split_cond": [], "default_left": [], "split_type": [0] * num_nodes,
             "categories": [[] for _ in range(num_nodes)]}
        for _i in range(num_nodes):
            _parent, cl, cr, sindex, info = struct.unpack_from("<iiiIf", blob, at)
            at += 20
            t["left"].append(cl)
            t["right"].append(cr)
            t["split_index"].append(sindex & 0x7fffffff)
            t["default_left"].append(sindex >> 31)
            t["split_cond"].append(info)
        at += 16 * num_nodes
        if leaf_vec:
            (n,) = struct.unpack_from("<Q", blob, at)
            at += 8 + 4 * n
        trees.append(t)
    info = struct.unpack_from("<%di" % num_trees, blob, at) if num_trees else ()
    if any(info):
        raise ValueError("multi-group models are not supported")
    return {"base_score": float(base_score), "trees": trees, "num_feature": int(num_feature), "objective": names[0]}


def parse_xgboost(blob: bytes) -> dict:
    k = 1
    while k < len(blob) and blob[k:k + 1] in (b" ", b"\n", b"\r", b"\t"):
        k += 1
    if blob[:1] != b"{":
        return parse_xgboost_legacy(blob)
    if blob[k:k + 1] in (b'"', b"}"):
        # float32 fields must go decimal -> f32 directly; keep the decimal text via parse_float
        doc = json.loads(blob.decode("utf-8"), parse_float=lambda s: _F(s))
    else:
        doc = parse_ubjson(blob)
    learner = doc["learner"]
    base_score = _num(learner["learner_model_param"]["base_score"])
    model = learner["gradient_booster"]["model"]
    trees = []
    for jt in model["trees"]:
        n = len(jt["left_children"])
        cats = [[] for _ in range(n)]
        if jt.get("categories_nodes"):
            for node, seg, size in zip(jt["categories_nodes"], jt["categories_segments"], jt["categories_sizes"]):
                cats[int(node)] = [int(c) for c in jt["categories"][int(seg):int(seg) + int(size)]]
        trees.append({
            "left": [int(x) for x in jt["left_children"]],
            "right": [int(x) for x in jt["right_children"]],
            "split_index": [int(x) for x in jt["split_indices"]],
            "split_cond": [x for x in jt["split_conditions"]],  # kept as-is; narrowed to f32 by the caller
            "default_left": [1 if x else 0 for x in jt["default_left"]],
            "split_type": [int(x) for x in jt.get("split_type", [0] * n)] or [0] * n,
            "categories": cats,
        })
    return {"base_score": base_score, "trees": trees,
            "num_feature": int(_num(learner["learner_model_param"].get("num_feature", 0)))}


class _F(float):
    """float that remembers its decimal text so f32 narrowing can be done from the text."""

    def __new__(cls, s: str):
        o = super().__new__(cls, s)
        o.text = s
        return o
