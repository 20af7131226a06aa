"""Forest decision rules pinned on an evaluator this repo did not write: scikit-learn's HistGradientBoostingRegressor.

tests/golden/make_hgb_golden.py fits two models (<= 16 leaves: the bit-vector scorer's range; 40 leaves: the tree walk),
with NaNs and two categorical columns, exports them into the on-disk formats the reference stores
(LambdaMARTRanker.scala:229-230 -> LightGBMBooster / XGBoostBooster bytes; build.sbt:57-58) and records what SKLEARN'S OWN
traversal (`_predictor.pyx _predict_from_raw_data`) returns.  What that pins, bit for bit, for oracle/ (CPU, here) and
for every HIP scorer variant (`-m gpu`, through the C ABI):

  * numerical  x <= threshold  goes left (LightGBM f64) /  x < up32(threshold)  (XGBoost f32);
  * NaN takes the node's own direction: decision_type bit 1 with missing_type NaN / `default_left`;
  * LightGBM's missing types None and Zero per column and its 1e-35 zero flush (the column rule applied to sklearn's input);
  * categorical bitsets: members go LEFT in LightGBM, members of `categories` go RIGHT in XGBoost;
  * a categorical node's NaN / negative / never-seen category goes RIGHT in LightGBM;
    an XGBoost categorical node sends NaN by default_left, never-seen categories LEFT unless listed, negatives LEFT;
  * summation in tree order: f64 from 0 (LightGBM), f32 from base_score (XGBoost).

The one thing the LightGBM FORMAT cannot say - "missing goes left" at a categorical node - is the documented divergence:
`test_lightgbm_cannot_express_categorical_missing_left` is a strict xfail over exactly the rows that meet such a node.
"""
import os

import numpy as np
import pytest

from oracle.forest import OracleForest

MODELS = ["hgb16", "hgb40"]


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    read = lambda f: open(os.path.join(golden_dir, f), "rb").read()  # noqa: E731
    return g, read(name + ".lgbm.txt"), read(name + "_mixed.lgbm.txt"), read(name + ".xgb.json")


def rows_of(g, group):
    return g["group"] == list(g["group_names"]).index(group)


# --------------------------------------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize("name", MODELS)
def test_oracle_lightgbm_nan_missing_type(golden_dir, name):
    g, lg, _, _ = load(golden_dir, name)
    got = OracleForest.from_lightgbm_text(lg).predict(g["X"])
    assert np.array_equal(got, g["expected_lgbm"])
    same = g["lgbm_same_as_fitted"]
    assert same[rows_of(g, "clean")].all()  # rows with known, present categories: the export IS the fitted model
    assert np.array_equal(got[same], g["expected_fitted"][same])


@pytest.mark.parametrize("name", MODELS)
def test_oracle_lightgbm_mixed_missing_types(golden_dir, name):
    g, _, mixed, _ = load(golden_dir, name)
    got = OracleForest.from_lightgbm_text(mixed).predict(g["X"])
    assert np.array_equal(got, g["expected_lgbm_mixed"])
    assert (g["expected_lgbm_mixed"] != g["expected_lgbm"]).sum() > 50  # the column rules do change routes


@pytest.mark.parametrize("name", MODELS)
def test_oracle_xgboost(golden_dir, name):
    g, _, _, xg = load(golden_dir, name)
    got = OracleForest.from_xgboost(xg).predict(g["X"])
    assert np.array_equal(got, g["expected_xgb_f32"])
    ok = g["xgb_same_as_fitted"]
    assert ok[~rows_of(g, "cat_negative")].all()  # NaN and never-seen categories ARE expressible in XGBoost's format
    assert np.array_equal(got[ok], g["expected_xgb_f32_fitted"][ok])


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.xfail(strict=True, reason="LightGBM's CategoricalDecision has no default direction: NaN / negative / never-seen "
                                       "categories always go right; sklearn's fitted model sends them left at these nodes")
def test_lightgbm_cannot_express_categorical_missing_left(golden_dir, name):
    g, lg, _, _ = load(golden_dir, name)
    div = ~g["lgbm_same_as_fitted"]
    assert div.sum() > 20
    got = OracleForest.from_lightgbm_text(lg).predict(g["X"][div])
    assert np.array_equal(got, g["expected_fitted"][div])


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.xfail(strict=True, reason="XGBoost's common::Decision sends an invalid (negative) category left whatever "
                                       "default_left says; sklearn treats it as missing")
def test_xgboost_cannot_express_negative_category_right(golden_dir, name):
    g, _, _, xg = load(golden_dir, name)
    div = ~g["xgb_same_as_fitted"]
    assert div.sum() > 20
    got = OracleForest.from_xgboost(xg).predict(g["X"][div])
    assert np.array_equal(got, g["expected_xgb_f32_fitted"][div])


@pytest.mark.parametrize("name", MODELS)
def test_xgboost_f32_sum_tracks_the_f64_model(golden_dir, name):
    """sanity of the export itself: base 0.5 + f32 leaves stay within 1e-4 of sklearn's f64 sum minus its baseline."""
    g, _, _, xg = load(golden_dir, name)
    ok = g["xgb_same_as_fitted"]
    got = OracleForest.from_xgboost(xg).predict(g["X"])
    assert np.allclose(got[ok] - 0.5, g["expected_fitted"][ok] - g["baseline"], atol=1e-4)


# --------------------------------------------------------------------------------------------- HIP scorers (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
def test_hip_scorers_against_sklearn(ctx, golden_dir, name):
    import metarank_amd as M
    from tests.test_score_gpu import _all_kernels, _predict_with
    g, lg, mixed, xg = load(golden_dir, name)
    X = g["X"]
    try:
        for blob, backend, key in ((lg, M.LIGHTGBM, "expected_lgbm"), (mixed, M.LIGHTGBM, "expected_lgbm_mixed"), (xg, M.XGBOOST, "expected_xgb_f32")):
            b = M.HipBooster(blob, backend, ctx)
            assert b.info()["bitvector"] == (1 if name == "hgb16" else 0)
            assert b.info()["n_categorical"] > 0
            for kernel, got in _all_kernels(b, X).items():
                assert np.array_equal(got, g[key]), (key, kernel)
            # one row at a time and in reverse: no dependence on the tile a row lands in
            assert np.array_equal(_predict_with(b, X[::-1].copy()), g[key][::-1])
            assert np.array_equal(np.concatenate([_predict_with(b, X[i:i + 1]) for i in range(0, len(X), 37)]), g[key][::37])
            b.close()
    finally:
        for k in ("MRK_SCORER", "MRK_QS_KERNEL", "MRK_QS_R", "MRK_QS_SPLIT", "MRK_WALK_TILE"):
            os.environ.pop(k, None)
        M.reload_switches()
