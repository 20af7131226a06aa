"""CPU: the model readers against files this repo did NOT write the way it usually does - the spellings other versions of the
libraries use: XGBoost `base_score` as a number, "5E-1", "[5E-1]", a one-element array; unknown keys and a different
key order; UBJSON; LightGBM headers with `feature_infos`, `tree_sizes`, CRLF line ends and the `parameters:` / `pandas_categorical`
trailers - and what the scorer does not implement must be MRK_ERR_UNSUPPORTED, never a silently different score.  Through the
host-only mrk_model_inspect (the reader + validator + packer of mrk_model_load, no device).
Reference: LambdaMARTRanker.scala:192-236 (the boosters come out of storage as bytes), build.sbt:57-58 (xgboost4j / lightgbm4j 4.6.0-1)."""
import copy
import ctypes as C
import json
import random

import numpy as np
import pytest

from metarank_amd import _native as N
from oracle.forest import OracleForest
from workloads import synth

Q = np.linspace(-2.0, 2.0, 33)[None, :].repeat(6, 0)


def inspect(backend: int, blob: bytes):
    info = N.mrk_model_info()
    rc = N.lib().mrk_model_inspect(backend, blob, len(blob), C.byref(info))
    return rc, info


def xgb_doc():
    return json.loads(synth.synthetic_xgb_model(n_trees=5, n_features=6, depth=3, quantiles=Q, cat_features=[2], cat_prob=0.2, fmt="json"))


@pytest.mark.parametrize("spelling, want", [(0.5, 0.5), ("5E-1", 0.5), ("[5E-1]", 0.5), ([0.5], 0.5), ("-1.25", -1.25), ("[1.17549435E-38]", float(np.float32(1.17549435e-38))),
                                            ("0.1", float(np.float32(0.1))), ("[3.4028235E38]", float(np.float32(3.4028235e38)))])
def test_xgboost_base_score_spellings(spelling, want):
    d = xgb_doc()
    d["learner"]["learner_model_param"]["base_score"] = spelling
    rc, info = inspect(1, json.dumps(d).encode())
    assert rc == 0, N.lib().mrk_last_error()
    assert info.base_score == want and info.n_trees == 5


@pytest.mark.parametrize("edit, status", [
    (lambda d: d["learner"]["learner_model_param"].__setitem__("base_score", "[5E-1,2.5E-1]"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["learner_model_param"].__setitem__("base_score", [0.5, 0.25]), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["learner_model_param"].__setitem__("num_target", "3"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["learner_model_param"].__setitem__("num_class", "3"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"].__setitem__("name", "dart"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"].__setitem__("weight_drop", [1.0] * 5), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"].__setitem__("name", "gblinear"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"]["model"].setdefault("gbtree_model_param", {}).__setitem__("num_parallel_tree", "4"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"]["model"].setdefault("gbtree_model_param", {}).__setitem__("size_leaf_vector", "2"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"]["model"].__setitem__("iteration_indptr", [0, 2, 4, 5]), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"]["model"].__setitem__("tree_info", [0, 1, 0, 1, 0]), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["gradient_booster"]["model"]["trees"][0].setdefault("tree_param", {}).__setitem__("size_leaf_vector", "3"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["objective"].__setitem__("name", "binary:logistic"), N.ERR_UNSUPPORTED),
    (lambda d: d["learner"]["learner_model_param"].__setitem__("base_score", "nan"), N.ERR_PARSE),
    (lambda d: d["learner"]["learner_model_param"].pop("base_score"), N.ERR_PARSE),
    (lambda d: d["learner"]["gradient_booster"]["model"]["trees"][1].pop("split_conditions"), N.ERR_PARSE),
])
def test_xgboost_models_the_scorer_does_not_implement_are_refused(edit, status):
    d = xgb_doc()
    edit(d)
    rc, _ = inspect(1, json.dumps(d).encode())
    assert rc == status, (rc, N.lib().mrk_last_error())


def _shuffled(v, rng, extra):
    """the same JSON document with every object's keys in another order and unknown keys sprinkled in"""
    if isinstance(v, dict):
        items = [(k, _shuffled(x, rng, extra)) for k, x in v.items()]
        for _ in range(rng.randrange(0, 3)):
            items.append((f"{rng.choice(extra)}_{rng.randrange(1000)}", rng.choice([None, 1, "x", [1, 2, {"a": []}], {"nested": {"deep": [0.5]}}, True])))
        rng.shuffle(items)
        return dict(items)
    if isinstance(v, list):
        return [_shuffled(x, rng, extra) for x in v]
    return v


def test_xgboost_key_order_and_unknown_keys_do_not_matter():
    """XGBoost's JSON writer emits keys in its own (version-dependent) order and every release adds fields: 40 reorderings of one
    model, with unknown keys at every level (also inside trees), predict like the original - compared through the oracle on the
    device-free side: same forest info, and the oracle's reader (oracle/formats.py) agrees on the scores."""
    d = xgb_doc()
    d["learner"]["feature_names"] = [f"f{i}" for i in range(6)]
    d["learner"]["feature_types"] = ["float", "float", "c", "float", "float", "float"]
    d["learner"]["attributes"] = {"best_iteration": "3", "scikit_learn": "{}"}
    d["learner"]["gradient_booster"]["model"]["iteration_indptr"] = list(range(6))
    d["learner"]["gradient_booster"]["model"]["gbtree_model_param"] = {"num_parallel_tree": "1", "num_trees": "5", "size_leaf_vector": "1"}
    d["version"] = [3, 0, 2]
    base = json.dumps(d).encode()
    rc0, i0 = inspect(1, base)
    assert rc0 == 0, N.lib().mrk_last_error()
    X = np.random.default_rng(3).normal(size=(200, 6)); X[:, 2] = np.random.default_rng(4).integers(0, 9, 200)
    want = OracleForest.from_xgboost(base).predict(X)
    rng = random.Random(11)
    for k in range(40):
        blob = json.dumps(_shuffled(d, rng, ["zz_new_field", "stats", "cats", "meta"])).encode()
        rc, i = inspect(1, blob)
        assert rc == 0, (k, N.lib().mrk_last_error())
        assert (i.n_trees, i.n_nodes, i.n_leaves, i.n_categorical, i.base_score, i.max_depth) == (i0.n_trees, i0.n_nodes, i0.n_leaves, i0.n_categorical, i0.base_score, i0.max_depth)
        np.testing.assert_array_equal(OracleForest.from_xgboost(blob).predict(X), want)


def lgbm_text():
    return synth.synthetic_lgbm_model(n_trees=6, n_features=6, num_leaves=16, max_depth=6, quantiles=Q, cat_features=[2], cat_prob=0.2, missing="per_feature").decode()


def test_lightgbm_text_as_the_library_writes_it():
    """What lib_lightgbm 4.x adds around the fields the scorer reads: `feature_infos`, `tree_sizes`, `monotone_constraints` in the
    header; `leaf_weight`, `leaf_count`, `internal_value`, `internal_weight`, `internal_count`, `shrinkage`, `is_linear=0` in every
    tree; after `end of trees` the `feature_importances:`, `parameters:` and `pandas_categorical:` sections; CRLF line ends."""
    t = lgbm_text()
    rc0, i0 = inspect(0, t.encode())
    assert rc0 == 0
    head, rest = t.split("\n\n", 1) if "\n\nTree=" in t else t.split("\nTree=", 1)
    lines = t.split("\n")
    out = []
    for ln in lines:
        out.append(ln)
        if ln.startswith("max_feature_idx="):
            out += ["feature_infos=[-2:2] [-2:2] 0:1:2:3:4:5:6:7:8 [-2:2] none [-2:2]", "monotone_constraints=0 0 0 0 0 0", "tree_sizes=" + " ".join(["1234"] * 6)]
        if ln.startswith("leaf_value="):
            n = len(ln.split("=")[1].split())
            out += ["leaf_weight=" + " ".join(["1.5"] * n), "leaf_count=" + " ".join(["10"] * n), "internal_value=" + " ".join(["0"] * max(n - 1, 1)),
                    "internal_weight=" + " ".join(["0"] * max(n - 1, 1)), "internal_count=" + " ".join(["20"] * max(n - 1, 1)), "is_linear=0", "shrinkage=0.1"]
    full = "\n".join(out)
    if "end of trees" not in full:
        full += "\nend of trees\n"
    full += ("\nfeature_importances:\nColumn_0=12\nColumn_3=7\n\nparameters:\n[boosting: gbdt]\n[objective: lambdarank]\n[num_leaves: 16]\n[learning_rate: 0.1]\nend of parameters\n\n"
             "pandas_categorical:[[\"a\", \"b\"]]\n")
    for blob in (full, full.replace("\n", "\r\n")):
        rc, i = inspect(0, blob.encode())
        assert rc == 0, N.lib().mrk_last_error()
        assert (i.n_trees, i.n_nodes, i.n_leaves, i.n_categorical, i.n_features) == (i0.n_trees, i0.n_nodes, i0.n_leaves, i0.n_categorical, i0.n_features)
    X = np.random.default_rng(5).normal(size=(100, 6)); X[:, 2] = np.random.default_rng(6).integers(0, 9, 100)
    np.testing.assert_array_equal(OracleForest.from_lightgbm_text(full.replace("\n", "\r\n").encode()).predict(X), OracleForest.from_lightgbm_text(t.encode()).predict(X))


def shuffled_lgbm(t: str, rng) -> str:
    """the same model string with the key=value lines of the header and of every tree block in another order, unknown keys added"""
    blocks = t.split("\n\n")
    out = []
    for blk in blocks:
        lines = blk.split("\n")
        if lines and (lines[0] == "tree" or lines[0].startswith("Tree=")):
            head, body = lines[:1], lines[1:]
            body += [f"zz_unknown_{rng.randrange(100)}=1 2 3", "another_new_field=abc"]
            rng.shuffle(body)
            out.append("\n".join(head + body))
        else:
            out.append(blk)
    return "\n\n".join(out)


def test_lightgbm_line_order_and_unknown_keys_do_not_matter():
    t = lgbm_text()
    rc0, i0 = inspect(0, t.encode())
    X = np.random.default_rng(8).normal(size=(200, 6)); X[:, 2] = np.random.default_rng(9).integers(0, 9, 200)
    want = OracleForest.from_lightgbm_text(t.encode()).predict(X)
    rng = random.Random(5)
    for k in range(40):
        blob = shuffled_lgbm(t, rng).encode()
        rc, i = inspect(0, blob)
        assert rc == 0, (k, N.lib().mrk_last_error())
        assert (i.n_trees, i.n_nodes, i.n_leaves, i.n_categorical, i.n_features, i.max_depth) == (i0.n_trees, i0.n_nodes, i0.n_leaves, i0.n_categorical, i0.n_features, i0.max_depth)
        np.testing.assert_array_equal(OracleForest.from_lightgbm_text(blob).predict(X), want)


@pytest.mark.parametrize("edit, status", [
    (lambda t: t.replace("objective=lambdarank", "objective=binary sigmoid:1"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("objective=lambdarank", "objective=multiclass num_class:3"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("objective=lambdarank", "objective=poisson"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("objective=lambdarank", "objective=regression sqrt"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("objective=lambdarank", "objective=regression"), 0),
    (lambda t: t.replace("objective=lambdarank", "objective=rank_xendcg"), 0),
    (lambda t: t.replace("objective=lambdarank", "objective=lambdarank\naverage_output"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("max_feature_idx=5", "max_feature_idx=5\nnum_class=3\nnum_tree_per_iteration=3"), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("is_linear=0", "is_linear=1", 1), N.ERR_UNSUPPORTED),
    (lambda t: t.replace("leaf_value=", "leaf_valu=", 1), N.ERR_PARSE),
])
def test_lightgbm_models_the_scorer_does_not_implement_are_refused(edit, status):
    t = edit(lgbm_text())
    rc, _ = inspect(0, t.encode())
    assert rc == status, (rc, N.lib().mrk_last_error())


def test_inspect_reports_what_load_would():
    blob = synth.synthetic_lgbm_model(n_trees=50, n_features=24)
    rc, i = inspect(0, blob)
    assert rc == 0 and i.n_trees == 50 and i.is_f64 == 1 and i.bitvector == 1 and i.device_bytes > 0 and i.tile_columns >= 24
    assert inspect(7, blob)[0] == N.ERR_INVALID_ARG
    assert inspect(0, b"not a model")[0] == N.ERR_PARSE
    assert N.lib().mrk_model_inspect(0, None, 0, None) == N.ERR_INVALID_ARG
