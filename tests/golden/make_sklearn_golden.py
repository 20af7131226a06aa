"""Generates tests/golden/sklearn_forest.npz (+ the two model files) — run once, committed.

scikit-learn's GradientBoostingRegressor is used as an INDEPENDENT tree evaluator: its trees are
exported into the two on-disk formats the reference stores (LightGBM model string, XGBoost JSON)
and its own predictions become the expected outputs.  This pins the format readers and the
traversal mechanics of oracle/ (and, through it, of the HIP scorer); it does not pin
LightGBM/XGBoost-specific rules (missing values, categorical splits), which sklearn does not share.

    python tests/golden/make_sklearn_golden.py
"""
import os
import sys

import numpy as np
from sklearn.ensemble import GradientBoostingRegressor

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from workloads import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def up32(thr: float) -> float:
    """smallest float32 strictly greater than thr: for float32 x,  x <= thr  <=>  x < up32(thr)."""
    t = np.float32(thr)
    if float(t) <= thr:
        t = np.nextafter(t, np.float32(np.inf))
    return float(t)


def main():
    rng = np.random.Generator(np.random.PCG64(20250718))
    n, d = 600, 10
    X = rng.normal(size=(n, d)).astype(np.float32)
    X[:, 3] = np.round(X[:, 3] * 2)  # ties / repeated values
    y = X[:, 0] * 2 + np.sin(X[:, 1] * 3) + (X[:, 2] > 0.3) * X[:, 4] + rng.normal(size=n) * 0.1
    lr = 0.1
    gbr = GradientBoostingRegressor(n_estimators=40, max_depth=4, learning_rate=lr, init="zero", random_state=7,
                                    subsample=0.8)
    gbr.fit(X, y)
    Xt = rng.normal(size=(257, d)).astype(np.float32)
    Xt[:, 3] = np.round(Xt[:, 3] * 2)
    # exact threshold hits
    t0 = gbr.estimators_[0, 0].tree_
    Xt[0, t0.feature[0]] = np.float32(t0.threshold[0])
    expected = gbr.predict(Xt)  # init='zero' => raw sum of lr * leaf in tree order, f64

    lgbm_trees, xgb_trees = [], []
    leaves_idx = np.zeros((len(Xt), len(gbr.estimators_)), dtype=np.int64)
    f32_sum = np.full(len(Xt), np.float32(0.5), dtype=np.float32)
    for ti, est in enumerate(gbr.estimators_[:, 0]):
        t = est.tree_
        nn = t.node_count
        inner = [i for i in range(nn) if t.children_left[i] != -1]
        leaf = [i for i in range(nn) if t.children_left[i] == -1]
        iid = {n_: k for k, n_ in enumerate(inner)}
        lid = {n_: k for k, n_ in enumerate(leaf)}
        ref = lambda c: iid[c] if c in iid else ~lid[c]
        lg = {"num_leaves": len(leaf), "leaf_value": [lr * float(t.value[i, 0, 0]) for i in leaf]}
        if inner:
            lg.update(split_feature=[int(t.feature[i]) for i in inner],
                      threshold=[float(t.threshold[i]) for i in inner],
                      decision_type=[(2 << 2) for _ in inner],  # missing type NaN, default right
                      left_child=[ref(int(t.children_left[i])) for i in inner],
                      right_child=[ref(int(t.children_right[i])) for i in inner])
        lgbm_trees.append(lg)
        xg = {
            "base_weights": [0.0] * nn, "categories": [], "categories_nodes": [], "categories_segments": [],
            "categories_sizes": [], "default_left": [0] * nn, "id": ti,
            "left_children": [int(c) for c in t.children_left], "loss_changes": [0.0] * nn,
            "parents": [2147483647] * nn, "right_children": [int(c) for c in t.children_right],
            "split_conditions": [up32(float(t.threshold[i])) if t.children_left[i] != -1
                                 else float(np.float32(lr * float(t.value[i, 0, 0]))) for i in range(nn)],
            "split_indices": [int(max(f, 0)) for f in t.feature], "split_type": [0] * nn, "sum_hessian": [1.0] * nn,
            "tree_param": {"num_deleted": "0", "num_feature": str(d), "num_nodes": str(nn), "size_leaf_vector": "1"},
        }
        xgb_trees.append(xg)
        app = est.apply(Xt)
        leaves_idx[:, ti] = app
        f32_sum = (f32_sum + np.array([np.float32(lr * float(t.value[a, 0, 0])) for a in app], dtype=np.float32)).astype(np.float32)

    lgbm_bytes = synth.write_lightgbm_text(lgbm_trees, d, objective="regression")
    xgb_bytes = synth.write_xgboost_json(synth.xgboost_document(xgb_trees, d, 0.5, "reg:squarederror"))
    open(os.path.join(HERE, "sklearn_forest.lgbm.txt"), "wb").write(lgbm_bytes)
    open(os.path.join(HERE, "sklearn_forest.xgb.json"), "wb").write(xgb_bytes)
    np.savez_compressed(os.path.join(HERE, "sklearn_forest.npz"), X=Xt.astype(np.float64), expected_f64=expected,
                        expected_xgb_f32=f32_sum.astype(np.float64))
    print("trees", len(lgbm_trees), "rows", len(Xt), "lgbm bytes", len(lgbm_bytes), "xgb bytes", len(xgb_bytes))


if __name__ == "__main__":
    main()
