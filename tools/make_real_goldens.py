#!/usr/bin/env python
"""Golden vectors from the REAL libraries behind the reference's scoring half - to be run on any box that has them
(`pip install lightgbm xgboost`, optionally `onnxruntime tokenizers` + the all-MiniLM-L6-v2 export); this build
environment has neither, nor a network.  Writes into tests/golden/real/:

    <name>.model   the model bytes exactly as the library serialises them (LightGBM model string; XGBoost JSON, UBJSON
                   and - where the installed version still writes it - the legacy binary format)
    <name>.npz     X (f64 rows, incl. the special values below) and the library's own predictions for them:
                   LightGBM  Booster.predict(X, raw_score=True)                       f64
                   XGBoost   Booster.predict(DMatrix(f32(X), missing=NaN), output_margin=True)   f32
                   - the calls ltrlib makes: LightGBMBooster.predictMat -> LGBM_BoosterPredictForMat(PREDICT_NORMAL; lambdarank
                   has no output transform), XGBoostBooster.predictMat -> DMatrix(float[], rows, cols, NaN) + predict
                   (ml/rank/LambdaMARTRanker.scala:347-359)
                   + importance_split / importance_gain (/ importance_total_gain): the library's feature importances
                   (LightGBM feature_importance, XGBoost get_score) = what ltrlib's Booster.weights() returns -> mrk_model_weights
    minilm.npz     (with --minilm-dir) the embeddings and cosines of OnnxBiencoderTest.scala:13-25 (0.539 / 0.738)

tests/test_real_goldens.py consumes whatever is there: the oracle (oracle/forest_oracle.cpp behind oracle/forest.py) and
the HIP path must reproduce the stored predictions bit for bit.  Models mirror what Metarank trains
(LambdaMARTRanker.scala:161-190): LightGBM lambdarank (numLeaves 16, maxDepth 8), XGBoost rank:ndcg with
tree_method=exact (maxDepth 6 / 8), a categorical feature, missing values.

    python tools/make_real_goldens.py [--out tests/golden/real] [--minilm-dir DIR]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dataset(rng, n_groups=120, per_group=30, d=12, cat_col=3, n_cats=12):
    n = n_groups * per_group
    X = rng.normal(size=(n, d))
    X[:, 1] = np.round(X[:, 1] * 3)                 # many ties, exact zeros
    X[:, 2] = np.abs(X[:, 2]) * (rng.random(n) > 0.4)  # zero-inflated
    X[:, cat_col] = rng.integers(0, n_cats, size=n)
    X[rng.random(X.shape) < 0.08] = np.nan
    X[np.isnan(X[:, cat_col]), cat_col] = 0
    w = rng.normal(size=d)
    rel = np.nan_to_num(X) @ w + (np.nan_to_num(X[:, cat_col]) % 3 == 0) * 1.5 + rng.normal(size=n) * 0.5
    y = np.clip(np.digitize(rel, np.quantile(rel, [0.5, 0.8, 0.95])), 0, 3).astype(int)
    return X, y, [per_group] * n_groups


def probe_rows(X, model_thresholds, cat_col):
    """rows that sit ON split thresholds, and the special values both libraries treat specially"""
    rng = np.random.default_rng(99)
    P = X[rng.integers(0, len(X), 400)].copy()
    for k, (f, t) in enumerate(model_thresholds[:200]):
        P[k % len(P), f] = t
    specials = [0.0, -0.0, 1e-36, -1e-36, 1e-35, 5e-324, np.nan, 1e30, -1e30, 3.4028234663852886e38]
    for k, v in enumerate(specials):
        P[200 + k, (k * 5) % P.shape[1]] = v
    P[230:240, cat_col] = [0, 1, 31, 32, 33, 63, 64, 1000, 1 << 20, 5]   # categories beyond the trained range
    P[240, cat_col] = -1
    P[241, cat_col] = np.nan
    P[242, cat_col] = 2.7        # LightGBM / XGBoost truncate
    return P


def make_lightgbm(out, rng):
    import lightgbm as lgb

    X, y, groups = dataset(rng)
    for name, params, cat in [
        ("lgbm_lambdarank", {}, [3]),
        ("lgbm_zero_as_missing", {"zero_as_missing": True}, [3]),
        ("lgbm_no_missing", {"use_missing": False}, []),
    ]:
        ds = lgb.Dataset(X, label=y, group=groups, categorical_feature=cat, free_raw_data=False)
        p = dict(objective="lambdarank", num_leaves=16, max_depth=8, learning_rate=0.1, min_data_in_leaf=5, verbose=-1, seed=1,
                 deterministic=True, num_threads=1)
        p.update(params)
        bst = lgb.train(p, ds, num_boost_round=60)
        model = bst.model_to_string()
        thr = []
        for t in bst.dump_model()["tree_info"]:
            st = [t["tree_structure"]]
            while st:
                nd = st.pop()
                if "split_feature" in nd:
                    if nd["decision_type"] == "<=":
                        thr.append((nd["split_feature"], float(nd["threshold"])))
                    st += [nd["left_child"], nd["right_child"]]
        P = probe_rows(X, thr, 3)
        P[250:252, 5] = [np.inf, -np.inf]
        pred = bst.predict(P, raw_score=True, num_threads=1)
        open(os.path.join(out, name + ".model"), "w").write(model)
        # Booster.weights() of ltrlib = LGBM_BoosterFeatureImportance: both types, for mrk_model_weights (include/mrk.h)
        np.savez_compressed(os.path.join(out, name + ".npz"), X=P, pred=pred.astype(np.float64), backend=0, library=f"lightgbm {lgb.__version__}",
                            importance_split=bst.feature_importance("split").astype(np.float64),
                            importance_gain=bst.feature_importance("gain").astype(np.float64))
        print(name, len(model), "bytes,", len(P), "rows")


def make_xgboost(out, rng):
    import xgboost as xgb

    X, y, groups = dataset(rng)
    Xf = X.astype(np.float32)
    for name, depth, cat in [("xgb_ndcg_d6", 6, False), ("xgb_ndcg_d8", 8, False), ("xgb_ndcg_cat", 6, True)]:
        ft = ["c" if (cat and j == 3) else "q" for j in range(X.shape[1])]
        dm = xgb.DMatrix(Xf, label=y, missing=np.nan, feature_types=ft, enable_categorical=cat)
        dm.set_group(groups)
        p = dict(objective="rank:ndcg", tree_method="exact" if not cat else "hist", max_depth=depth, eta=0.1, seed=1, nthread=1,
                 max_cat_to_onehot=1)
        bst = xgb.train(p, dm, num_boost_round=50)
        thr = []
        for t in bst.get_dump(dump_format="json"):
            import json
            st = [json.loads(t)]
            while st:
                nd = st.pop()
                if "split" in nd and "split_condition" in nd and not isinstance(nd["split_condition"], list):
                    thr.append((int(str(nd["split"]).lstrip("f")), float(nd["split_condition"])))
                st += nd.get("children", [])
        P = probe_rows(X, thr, 3)
        Pf = P.astype(np.float32)
        pred = bst.predict(xgb.DMatrix(Pf, missing=np.nan, feature_types=ft, enable_categorical=cat), output_margin=True)
        forms = {"json": bytes(bst.save_raw(raw_format="json")), "ubj": bytes(bst.save_raw(raw_format="ubj"))}
        if not cat:
            try:
                forms["legacy"] = bytes(bst.save_raw(raw_format="deprecated"))   # xgboost < 3: the pre-1.0 binary layout
            except Exception as e:  # noqa: BLE001
                print("  (no legacy binary from this xgboost version:", e, ")")
        def score(kind):   # Booster.getScore(featureMap, kind): features never split on are absent -> 0
            got = bst.get_score(importance_type=kind)
            return np.array([float(got.get(f"f{j}", 0.0)) for j in range(X.shape[1])], dtype=np.float64)
        imp = {"importance_split": score("weight"), "importance_gain": score("gain"), "importance_total_gain": score("total_gain")}
        for fmt, blob in forms.items():
            open(os.path.join(out, f"{name}.{fmt}.model"), "wb").write(blob)
            np.savez_compressed(os.path.join(out, f"{name}.{fmt}.npz"), X=P, pred=pred.astype(np.float32), backend=1,
                                library=f"xgboost {xgb.__version__}", **imp)
            print(name, fmt, len(blob), "bytes,", len(P), "rows")


def make_minilm(out, d):
    """OnnxBiencoderTest.scala:13-25 with onnxruntime + tokenizers on the reference's own export
    (https://huggingface.co/metarank/all-MiniLM-L6-v2: pytorch_model.onnx + tokenizer.json, copied into `d`)."""
    import onnxruntime as ort
    from tokenizers import Tokenizer

    tok = Tokenizer.from_file(os.path.join(d, "tokenizer.json"))
    tok.enable_padding()
    tok.enable_truncation(max_length=512)
    sess = ort.InferenceSession(os.path.join(d, "pytorch_model.onnx"))
    texts = ["copper frying pan", "How many people live in Berlin?", "Berlin is well known for its museums.",
             "Berlin had a population of 3,520,031 registered inhabitants in an area of 891.82 square kilometers."]
    embs = []
    for batch in (texts[:1], texts[1:]):   # the test embeds them in two calls (padding differs)
        enc = tok.encode_batch(batch)
        ids = np.array([e.ids for e in enc], dtype=np.int64)
        mask = np.array([e.attention_mask for e in enc], dtype=np.int64)
        types = np.array([e.type_ids for e in enc], dtype=np.int64)
        names = [i.name for i in sess.get_inputs()]
        feed = {"input_ids": ids, "attention_mask": mask, "token_type_ids": types}
        hidden = sess.run(None, {k: v for k, v in feed.items() if k in names})[0]
        for h, m in zip(hidden, mask):     # OnnxBiEncoder.avgpool: f64 sum over the first sum(mask) tokens -> f32
            n = int(m.sum())
            embs.append((h[:n].astype(np.float64).sum(axis=0) / n).astype(np.float32))
    embs = np.stack(embs)

    def cos(a, b):  # DistanceFunction.scala:14-26: f64 accumulators, f32 x f64 products
        a, b = a.astype(np.float32), b.astype(np.float32).astype(np.float64)
        return float((a.astype(np.float64) * b).sum() / (np.sqrt((a * a).astype(np.float64).sum()) * np.sqrt((b * b).sum())))
    d1, d2 = cos(embs[1], embs[2]), cos(embs[1], embs[3])
    print("cosines", d1, d2, "(OnnxBiencoderTest expects 0.539 / 0.738 +- 1e-3)")
    np.savez_compressed(os.path.join(out, "minilm.npz"), texts=np.array(texts), embeddings=embs, d1=d1, d2=d2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "real"))
    ap.add_argument("--minilm-dir", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rng = np.random.default_rng(20250718)
    done = []
    for lib, fn in (("lightgbm", make_lightgbm), ("xgboost", make_xgboost)):
        try:
            __import__(lib)
        except ImportError:
            print(f"{lib} is not installed: skipped", file=sys.stderr)
            continue
        fn(a.out, rng)
        done.append(lib)
    if a.minilm_dir:
        make_minilm(a.out, a.minilm_dir)
        done.append("minilm")
    if not done:
        sys.exit("nothing to do: install lightgbm and / or xgboost (see the docstring)")


if __name__ == "__main__":
    main()
