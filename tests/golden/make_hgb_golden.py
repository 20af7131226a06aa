"""Generates tests/golden/hgb_*.{npz,lgbm.txt,xgb.json} - run once, committed.

    python tests/golden/make_hgb_golden.py

scikit-learn's HistGradientBoostingRegressor (1.7, in this image) is a LightGBM work-alike with the rules the
GradientBoostingRegressor goldens (make_sklearn_golden.py) cannot pin: a per-node missing direction and categorical
bitsets.  It is used here as an INDEPENDENT evaluator - the traversal that produces every expected value is sklearn's
own Cython `_predict_from_raw_data` (ensemble/_hist_gradient_boosting/_predictor.pyx:37-88):

    NaN                      -> node.missing_go_to_left
    categorical, value < 0   -> treated as missing
    categorical, in left set -> left;  known category not in it -> right;  unknown category -> treated as missing
    numerical                -> left iff  x <= num_threshold  (f64)

Fitted trees are exported into the two on-disk formats the reference stores (LambdaMARTRanker.scala:229-230 hands the
bytes to LightGBMBooster / XGBoostBooster; build.sbt:57-58 pins ltrlib 0.2.6 / lightgbm4j 4.6.0-1):

  LightGBM text   numerical node: decision_type = missing_type NaN | default_left (= missing_go_to_left), threshold as is;
                  categorical node: cat_boundaries / cat_threshold = the node's raw left bitset.  The format has NO
                  missing direction for a categorical node: NaN, negative and never-seen categories go RIGHT
                  (tree.h CategoricalDecision).  The expected values of the LightGBM files therefore come from sklearn
                  evaluating a TWIN of the model whose categorical nodes have missing_go_to_left = 0 - on rows that
                  never meet such a node the twin and the fitted model agree bit for bit (`lgbm_same_as_fitted`),
                  the others are kept as the documented divergence (`expected_fitted` holds the fitted model's values).
                  The baseline (sklearn's mean of y) is a leading one-leaf tree, so  0 + b + t1 + t2 ...  are the very
                  additions of `_raw_predict`.
  LightGBM, mixed missing types (hgb16_mixed): the same trees with a missing type per numerical COLUMN (None / Zero /
                  NaN, as LightGBM's bin mappers assign them).  sklearn has only the NaN rule, so the column rules are
                  applied to the INPUT and sklearn evaluates the result: None = "NaN reads as 0.0", Zero = "0.0 (and NaN,
                  which reads as 0.0) is missing", every column = "|x| <= 1e-35 reads as 0.0" (c_api dense rows).
  XGBoost JSON    numerical: split_condition = the smallest f32 above the threshold (x <= t  <=>  x < up32(t) for f32 x),
                  default_left = missing_go_to_left; categorical (split_type 1): members of `categories` go RIGHT, so the
                  set is the complement of the left set - over the known categories when missing goes left (a never-seen
                  category must go left), over all of [0, 256) when it goes right (a never-seen category must go right).
                  A NEGATIVE category is invalid and goes left whatever the node says (common::Decision): rows with one
                  take their expected values from the twin whose categorical nodes have missing_go_to_left = 1.
                  Leaves narrowed to f32; expected = f32 additions in tree order from base_score 0.5.

All test values are float32-representable (ltrlib narrows to float for XGBoost).
"""
import json
import os
import sys

import numpy as np
from sklearn.ensemble import HistGradientBoostingRegressor
from sklearn.ensemble._hist_gradient_boosting.predictor import TreePredictor

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from workloads import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
D = 10
CATS = (40, 12)           # column 0: 40 categories, never NaN in training; column 1: 12 categories, NaN in training
MT_NONE, MT_ZERO, MT_NAN = 0, 1, 2
K_ZERO = float(np.float32(1e-35))   # include/LightGBM/meta.h: const double kZeroThreshold = 1e-35f  (the FLOAT literal, widened)


def up32(thr: float) -> float:
    """smallest float32 strictly greater than thr: for float32 x,  x <= thr  <=>  x < up32(thr)."""
    if thr >= 3.4028234663852886e38:
        return 3.4028234663852886e38
    t = np.float32(thr)
    if float(t) <= thr:
        t = np.nextafter(t, np.float32(np.inf))
    return float(t)


def make_data(rng, n, train):
    X = rng.normal(size=(n, D)).astype(np.float32).astype(np.float64)
    X[:, 0] = rng.integers(0, CATS[0], size=n)
    X[:, 1] = rng.integers(0, CATS[1], size=n)
    X[:, 4] = np.round(X[:, 4] * 2)                                    # ties, exact zeros
    X[rng.random(n) < 0.2, 3] = 0.0                                    # a column with many exact zeros
    m = rng.random((n, D)) < 0.08
    m[:, 0] = False
    X[m] = np.nan
    if train:                                                          # every category present: the encoder is the identity
        X[:CATS[0], 0] = np.arange(CATS[0])
        X[:CATS[1], 1] = np.arange(CATS[1])
    return X


def target(rng, X):
    z = np.nan_to_num(X)
    return (z[:, 2] * 2 + np.sin(z[:, 5] * 3) + np.isin(X[:, 0], [3, 7, 9, 22, 38]) * 1.5 + np.isin(X[:, 1], [1, 2, 5]) * z[:, 6]
            + np.isnan(X[:, 1]) * 2.0 + np.isnan(X[:, 7]) * 1.0 - np.isnan(X[:, 8]) * z[:, 9] + (z[:, 3] == 0) * 0.7
            + rng.normal(size=len(X)) * 0.1)


def test_rows(rng, model):
    """name -> rows; every value float32-representable."""
    groups = {}
    X = make_data(rng, 240, False)
    X[:, 1] = np.where(np.isnan(X[:, 1]), 3.0, X[:, 1])                # 'clean': categorical cells known and present
    tiny = [float(np.float32(v)) for v in (1e-36, -1e-36, 1e-35, 2e-35, 1e30)]   # f32(1e-35) IS LightGBM's kZeroThreshold (1e-35f)
    X[5, 2:] = [0.0, -0.0, tiny[0], tiny[1], tiny[2], tiny[3], np.nan, tiny[4]]
    X[6, 2:] = [tiny[2], tiny[3], 0.0, tiny[0], -0.0, tiny[1], tiny[0], np.nan]     # zeros of every kind in the numerical columns
    k = 10
    for it in model._predictors[:12]:                                  # exact threshold hits (f32 neighbours of the f64 threshold)
        nd = it[0].nodes
        for node in nd[(nd["is_leaf"] == 0) & (nd["is_categorical"] == 0)][:3]:
            t = np.float32(node["num_threshold"])
            if not np.isfinite(t):
                continue
            for v in (t, np.nextafter(t, np.float32(-np.inf)), np.nextafter(t, np.float32(np.inf))):
                X[k, node["feature_idx"]] = float(v)
                k += 1
    assert k < 240
    groups["clean"] = X
    X = make_data(rng, 80, False)
    X[:40, 0] = np.nan
    X[20:60, 1] = np.nan
    groups["cat_nan"] = X
    X = make_data(rng, 80, False)
    X[:, 1] = np.where(np.isnan(X[:, 1]), 3.0, X[:, 1])
    X[:40, 0] = rng.integers(CATS[0], 200, size=40)
    X[20:60, 1] = rng.integers(CATS[1], 200, size=40)
    X[60:, 1] = rng.integers(CATS[1], 32, size=20)                     # never seen, but inside the bitset's first word
    groups["cat_unknown"] = X
    X = make_data(rng, 80, False)
    X[:, 1] = np.where(np.isnan(X[:, 1]), 3.0, X[:, 1])
    X[:40, 0] = -rng.integers(1, 5, size=40)
    X[20:60, 1] = -rng.integers(1, 5, size=40)
    X[60:, 0] = -1.0
    groups["cat_negative"] = X
    return groups


def twin(model, cat_missing_left):
    out = []
    for it in model._predictors:
        p = it[0]
        nodes = p.nodes.copy()
        cat = (nodes["is_categorical"] == 1) & (nodes["is_leaf"] == 0)
        nodes["missing_go_to_left"][cat] = cat_missing_left
        out.append([TreePredictor(nodes, p.binned_left_cat_bitsets, p.raw_left_cat_bitsets)])
    return out


def raw_predict(model, predictors, X):
    """HistGradientBoostingRegressor._raw_predict (gradient_boosting.py) over another predictor list: zeros + baseline,
    then += each tree's prediction, all by sklearn's own code."""
    raw = np.zeros((len(X), 1), dtype=model._baseline_prediction.dtype, order="F")
    raw += model._baseline_prediction
    model._predict_iterations(np.ascontiguousarray(X), predictors, raw, False, 1)
    return raw[:, 0].copy()


def per_tree(model, predictors, X):
    kb, fmap = model._bin_mapper.make_known_categories_bitsets()
    return np.stack([it[0].predict(np.ascontiguousarray(X), kb, fmap, 1) for it in predictors], axis=1)


def bits_of(words):
    return [w * 32 + b for w in range(len(words)) for b in range(32) if (int(words[w]) >> b) & 1]


def export_lgbm(model, missing_types):
    """missing_types[j] for numerical column j (categorical nodes carry MT_NAN, which CategoricalDecision ignores)."""
    trees = [{"num_leaves": 1, "leaf_value": [float(model._baseline_prediction[0, 0])]}]
    for it in model._predictors:
        p = it[0]
        nd = p.nodes
        inner = [i for i in range(len(nd)) if not nd["is_leaf"][i]]
        leaf = [i for i in range(len(nd)) if nd["is_leaf"][i]]
        iid = {n_: k for k, n_ in enumerate(inner)}
        lid = {n_: k for k, n_ in enumerate(leaf)}
        ref = lambda c: iid[c] if c in iid else ~lid[c]  # noqa: E731
        t = {"num_leaves": len(leaf), "leaf_value": [float(nd["value"][i]) for i in leaf]}
        if inner:
            thr, dt, cb, ct = [], [], [0], []
            for i in inner:
                f = int(nd["feature_idx"][i])
                if nd["is_categorical"][i]:
                    words = [int(w) for w in p.raw_left_cat_bitsets[nd["bitset_idx"][i]]]
                    while len(words) > 1 and words[-1] == 0:
                        words.pop()
                    thr.append(float(len(cb) - 1))
                    ct += words
                    cb.append(len(ct))
                    dt.append(1 | (MT_NAN << 2))
                else:
                    x = float(nd["num_threshold"][i])
                    thr.append(x if np.isfinite(x) else 1e300)
                    dt.append((int(missing_types[f]) << 2) | (2 if nd["missing_go_to_left"][i] else 0))
            t.update(split_feature=[int(nd["feature_idx"][i]) for i in inner], threshold=thr, decision_type=dt,
                     left_child=[ref(int(nd["left"][i])) for i in inner], right_child=[ref(int(nd["right"][i])) for i in inner])
            if len(cb) > 1:
                t.update(cat_boundaries=cb, cat_threshold=ct)
        trees.append(t)
    return synth.write_lightgbm_text(trees, D, objective="regression")


def export_xgb(model):
    kb, fmap = model._bin_mapper.make_known_categories_bitsets()
    known = {f: set(bits_of(kb[fmap[f]])) for f in range(D) if model.is_categorical_[f]}
    trees = []
    for ti, it in enumerate(model._predictors):
        p = it[0]
        nd = p.nodes
        nn = len(nd)
        cats, cat_nodes, segs, sizes = [], [], [], []
        sc, st = [], []
        for i in range(nn):
            if nd["is_leaf"][i]:
                sc.append(float(np.float32(nd["value"][i])))
                st.append(0)
            elif nd["is_categorical"][i]:
                left = set(bits_of(p.raw_left_cat_bitsets[nd["bitset_idx"][i]]))
                universe = known[int(nd["feature_idx"][i])] if nd["missing_go_to_left"][i] else set(range(256))
                right = sorted(universe - left)
                cat_nodes.append(i)
                segs.append(len(cats))
                sizes.append(len(right))
                cats += right
                sc.append(0.0)
                st.append(1)
            else:
                sc.append(up32(float(nd["num_threshold"][i])))
                st.append(0)
        trees.append({
            "base_weights": [0.0] * nn, "categories": cats, "categories_nodes": cat_nodes, "categories_segments": segs,
            "categories_sizes": sizes, "default_left": [int(nd["missing_go_to_left"][i]) if not nd["is_leaf"][i] else 0 for i in range(nn)],
            "id": ti, "left_children": [int(nd["left"][i]) if not nd["is_leaf"][i] else -1 for i in range(nn)],
            "loss_changes": [0.0] * nn, "parents": [2147483647] * nn,
            "right_children": [int(nd["right"][i]) if not nd["is_leaf"][i] else -1 for i in range(nn)],
            "split_conditions": sc, "split_indices": [int(nd["feature_idx"][i]) if not nd["is_leaf"][i] else 0 for i in range(nn)],
            "split_type": st, "sum_hessian": [1.0] * nn,
            "tree_param": {"num_deleted": "0", "num_feature": str(D), "num_nodes": str(nn), "size_leaf_vector": "1"},
        })
    return synth.write_xgboost_json(synth.xgboost_document(trees, D, 0.5, "reg:squarederror"))


def f32_sum(leaves):
    acc = np.full(leaves.shape[0], np.float32(0.5), dtype=np.float32)
    for t in range(leaves.shape[1]):
        acc = (acc + leaves[:, t].astype(np.float32)).astype(np.float32)
    return acc.astype(np.float64)


def apply_missing_types(X, missing_types):
    """the column rules of LightGBM's None / Zero missing types and of its dense-row reader, applied to the input so
    that sklearn's NaN rule evaluates them."""
    Z = X.copy()
    for j in range(D):
        col = Z[:, j]
        tiny = np.abs(col) <= K_ZERO                  # c_api RowFunctionFromDenseMatric: not copied, reads back 0.0
        col[tiny] = 0.0
        if j < 2:
            continue                                  # categorical columns: CategoricalDecision has no missing type
        if missing_types[j] == MT_NONE:
            col[np.isnan(col)] = 0.0                  # NaN reads as 0.0 and is compared
        elif missing_types[j] == MT_ZERO:
            col[np.isnan(col)] = 0.0
            col[col == 0.0] = np.nan                  # zero (incl. a NaN read as zero) takes the default direction
    return Z


def build(name, seed, **hgb):
    rng = np.random.Generator(np.random.PCG64(seed))
    Xtr = make_data(rng, 4000, True)
    model = HistGradientBoostingRegressor(categorical_features=[0, 1], early_stopping=False, random_state=seed, **hgb)
    model.fit(Xtr, target(rng, Xtr))
    groups = test_rows(rng, model)
    X = np.concatenate(list(groups.values()))
    group_of = np.concatenate([np.full(len(v), i) for i, v in enumerate(groups.values())])
    assert np.array_equal(X.astype(np.float32).astype(np.float64), X, equal_nan=True)
    clean = groups["clean"]
    assert np.array_equal(model._preprocessor.transform(clean), clean, equal_nan=True)  # identity encoder, identity column order

    fitted = model._predictors
    expected_fitted = model._raw_predict(X)[:, 0]                                          # sklearn's public path (encoder + trees)
    assert np.array_equal(raw_predict(model, fitted, X), expected_fitted)                  # == the trees over the raw cells
    t0, t1 = twin(model, 0), twin(model, 1)
    expected_lgbm = raw_predict(model, t0, X)
    same = expected_lgbm == expected_fitted
    neg = group_of == list(groups).index("cat_negative")
    leaves_fitted = per_tree(model, fitted, X)
    leaves = np.where(neg[:, None], per_tree(model, t1, X), leaves_fitted)
    expected_xgb = f32_sum(leaves)
    xgb_same = (leaves == leaves_fitted).all(axis=1)

    mts = np.array([MT_NAN, MT_NAN] + [(MT_NONE, MT_ZERO, MT_NAN)[(j + seed) % 3] for j in range(2, D)])
    mts[3] = MT_ZERO                                                                       # the column full of exact zeros
    expected_mixed = raw_predict(model, t0, apply_missing_types(X, mts))

    open(os.path.join(HERE, f"{name}.lgbm.txt"), "wb").write(export_lgbm(model, [MT_NAN] * D))
    open(os.path.join(HERE, f"{name}_mixed.lgbm.txt"), "wb").write(export_lgbm(model, mts))
    open(os.path.join(HERE, f"{name}.xgb.json"), "wb").write(export_xgb(model))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), X=X, group=group_of, group_names=np.array(list(groups)),
                        expected_fitted=expected_fitted, expected_lgbm=expected_lgbm, lgbm_same_as_fitted=same,
                        expected_xgb_f32=expected_xgb, expected_xgb_f32_fitted=f32_sum(leaves_fitted), xgb_same_as_fitted=xgb_same,
                        baseline=float(model._baseline_prediction[0, 0]),
                        expected_lgbm_mixed=expected_mixed, missing_types=mts)
    n_cat = sum(int(((it[0].nodes["is_categorical"] == 1) & (it[0].nodes["is_leaf"] == 0)).sum()) for it in fitted)
    n_cat_left = sum(int(((it[0].nodes["is_categorical"] == 1) & (it[0].nodes["missing_go_to_left"] == 1)).sum()) for it in fitted)
    n_num_left = sum(int(((it[0].nodes["is_categorical"] == 0) & (it[0].nodes["is_leaf"] == 0) & (it[0].nodes["missing_go_to_left"] == 1)).sum()) for it in fitted)
    n_inner = sum(int((it[0].nodes["is_leaf"] == 0).sum()) for it in fitted)
    summary = {"trees": len(fitted), "max_leaves": max(it[0].get_n_leaf_nodes() for it in fitted), "inner_nodes": n_inner,
               "categorical_nodes": n_cat, "categorical_missing_left": n_cat_left, "numerical_missing_left": n_num_left,
               "rows": len(X), "rows_lgbm_diverges_from_fitted": int((~same).sum()),
               "rows_xgb_diverges_from_fitted": int((~xgb_same).sum()),
               "rows_mixed_differs_from_nan_only": int((expected_mixed != expected_lgbm).sum())}
    print(name, json.dumps(summary))
    return summary


def main():
    out = {"hgb16": build("hgb16", 20250718, max_iter=60, max_leaf_nodes=16, max_depth=8, learning_rate=0.1, min_samples_leaf=5),
           "hgb40": build("hgb40", 20250719, max_iter=25, max_leaf_nodes=40, max_depth=None, learning_rate=0.15, min_samples_leaf=3)}
    json.dump(out, open(os.path.join(HERE, "hgb_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
