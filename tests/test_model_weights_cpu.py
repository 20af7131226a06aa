"""Booster.weights() (ltrlib; reference call site ml/rank/LambdaMARTRanker.scala:391-406) = mrk_model_weights /
mrk_model_inspect_weights.  The formulas follow the libraries' published sources (LightGBM GBDT::FeatureImportance, XGBoost
GBTree::FeatureScore - neither library is in the image, include/mrk.h says so); the checker here is an independent numpy
restatement over the tree dictionaries the model files are WRITTEN from, so what is pinned is: the readers keep every gain, in
every format, against the node renumbering the loaders do; the per-library arithmetic (double sum of positive float gains /
float sum and float quotient); the shape contract of the reference's `w(offset)` / `w.slice(offset, offset + size)`."""
import numpy as np
import pytest

from metarank_amd import _native as N
from metarank_amd.booster import LIGHTGBM, XGBOOST, inspect_weights
from workloads import synth

SPLIT, GAIN, TOTAL = 0, 1, 2


def lgbm_model(seed, n_trees=40, n_features=9):
    rng = np.random.default_rng(seed)
    trees = []
    for _ in range(n_trees):
        t = synth.random_lgbm_tree(rng, n_features, num_leaves=int(rng.integers(1, 17)), cat_features=[3], cat_prob=0.1)
        if t["num_leaves"] > 1:
            g = np.abs(rng.normal(size=len(t["split_feature"])) * 50).astype(np.float32)
            g[rng.random(len(g)) < 0.15] = 0.0          # LightGBM skips splits whose gain is not > 0
            g[rng.random(len(g)) < 0.05] = -1.5
            t["split_gain"] = [float(x) for x in g]
        trees.append(t)
    return trees, n_features


def lgbm_expected(trees, n, kind):
    out = np.zeros(n, dtype=np.float64)
    for t in trees:
        for f, g in zip(t.get("split_feature", []), t.get("split_gain", [])):
            g = np.float32(g)
            if g > 0:
                out[f] += 1.0 if kind == SPLIT else np.float64(g)
    return out


def xgb_model(seed, n_trees=30, n_features=11, cat=True):
    rng = np.random.default_rng(seed)
    trees = []
    for _ in range(n_trees):
        t = synth.random_xgb_tree(rng, n_features, depth=int(rng.integers(1, 5)), complete=False, cat_features=[2] if cat else None,
                                  cat_prob=0.1 if cat else 0.0)
        t["loss_changes"] = [float(np.float32(abs(rng.normal()) * 30)) if l != -1 else 0.0 for l in t["left_children"]]
        trees.append(t)
    return trees, n_features


def xgb_expected(trees, n, kind):
    total = np.zeros(n, dtype=np.float32)
    count = np.zeros(n, dtype=np.int64)
    for t in trees:   # RegTree::WalkTree: a stack walk that pushes the left child, then the right one
        stack = [0]
        while stack:
            u = stack.pop()
            if t["left_children"][u] == -1:
                continue
            f = t["split_indices"][u]
            count[f] += 1
            total[f] = np.float32(total[f] + np.float32(t["loss_changes"][u]))
            stack.append(t["left_children"][u])
            stack.append(t["right_children"][u])
    out = np.zeros(n, dtype=np.float64)
    for i in range(n):
        if count[i]:
            out[i] = count[i] if kind == SPLIT else total[i] if kind == TOTAL else np.float32(total[i] / np.float32(count[i]))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lightgbm_importance(seed):
    trees, n = lgbm_model(seed)
    blob = synth.write_lightgbm_text(trees, n)
    for kind in (SPLIT, GAIN, TOTAL):
        got = inspect_weights(blob, LIGHTGBM, n, kind)
        assert np.array_equal(got, lgbm_expected(trees, n, kind)), kind   # bit for bit: same additions in the same order
    assert np.array_equal(inspect_weights(blob, LIGHTGBM, n, GAIN), inspect_weights(blob, LIGHTGBM, n, TOTAL))
    assert inspect_weights(blob, LIGHTGBM, n, GAIN).sum() > 0


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("fmt", ["json", "ubj", "legacy", "binf"])
def test_xgboost_importance_in_every_serialisation(seed, fmt):
    trees, n = xgb_model(seed, cat=fmt in ("json", "ubj"))
    doc = synth.xgboost_document(trees, n)
    blob = {"json": synth.write_xgboost_json, "ubj": synth.write_xgboost_ubjson, "legacy": synth.write_xgboost_legacy,
            "binf": lambda d: synth.write_xgboost_legacy(d, binf=True)}[fmt](doc)
    for kind in (SPLIT, GAIN, TOTAL):
        assert np.array_equal(inspect_weights(blob, XGBOOST, n, kind), xgb_expected(trees, n, kind)), (fmt, kind)


def test_shape_contract_of_the_reference_descriptor():
    """LambdaMARTRanker.scala:391-406 indexes w by DatasetDescriptor offsets: one entry per matrix column; columns the forest
    never saw are 0.0; an array shorter than the model's feature count would be an IndexOutOfBounds on the JVM."""
    trees, n = lgbm_model(7)
    blob = synth.write_lightgbm_text(trees, n)
    wide = inspect_weights(blob, LIGHTGBM, n + 5, GAIN)
    assert np.array_equal(wide[:n], lgbm_expected(trees, n, GAIN)) and not wide[n:].any()
    with pytest.raises(N.MrkError) as e:
        inspect_weights(blob, LIGHTGBM, n - 1, GAIN)
    assert e.value.status == N.ERR_DIM_MISMATCH
    with pytest.raises(N.MrkError) as e:
        inspect_weights(blob, LIGHTGBM, n, 3)
    assert e.value.status == N.ERR_INVALID_ARG
    with pytest.raises(N.MrkError) as e:
        inspect_weights(b"not a model", XGBOOST, n, GAIN)
    assert e.value.status == N.ERR_PARSE


def test_a_file_without_gains_answers_counts_only():
    trees, n = xgb_model(5, cat=False)
    want = xgb_expected(trees, n, SPLIT)
    doc = synth.xgboost_document(trees, n)
    for t in doc["learner"]["gradient_booster"]["model"]["trees"]:
        del t["loss_changes"]
    blob = synth.write_xgboost_json(doc)
    assert np.array_equal(inspect_weights(blob, XGBOOST, n, SPLIT), want)
    with pytest.raises(N.MrkError) as e:
        inspect_weights(blob, XGBOOST, n, GAIN)
    assert e.value.status == N.ERR_UNSUPPORTED
    text = synth.write_lightgbm_text(lgbm_model(5)[0], 9).decode()
    text = "\n".join(l for l in text.split("\n") if not l.startswith("split_gain="))
    with pytest.raises(N.MrkError) as e:
        inspect_weights(text.encode(), LIGHTGBM, 9, GAIN)
    assert e.value.status == N.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_handle_and_bytes_agree_on_the_device():
    import metarank_amd as M
    ctx = M.Context(0)
    trees, n = lgbm_model(11)
    blob = synth.write_lightgbm_text(trees, n)
    b = M.HipBooster(blob, M.LIGHTGBM, ctx)
    assert np.array_equal(b.weights(), lgbm_expected(trees, n, GAIN))
    assert np.array_equal(b.weights(n + 3, SPLIT)[:n], lgbm_expected(trees, n, SPLIT))
    xt, xn = xgb_model(4)
    xb = M.HipBooster(synth.write_xgboost_json(synth.xgboost_document(xt, xn)), M.XGBOOST, ctx)
    assert np.array_equal(xb.weights(), xgb_expected(xt, xn, GAIN))
