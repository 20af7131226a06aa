"""GPU parity: the HIP forest scorer (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): identical ordering, |score delta| <= 1e-5; the kernel adds leaves in
tree order in the library's own precision, so these tests demand bit-exact scores.
"""
import math
import os

import numpy as np
import pytest

import metarank_amd as M
from metarank_amd import synth
from oracle.forest import OracleForest

pytestmark = pytest.mark.gpu
NAN = float("nan")


def make_X(rng, rows, cols, cat_col=None, n_cats=16, nan_frac=0.05, zero_frac=0.05):
    X = rng.normal(size=(rows, cols))
    if cat_col is not None:
        X[:, cat_col] = rng.integers(-1, n_cats + 2, size=rows)
    m = rng.random(X.shape)
    X[m < nan_frac] = NAN
    X[(m >= nan_frac) & (m < nan_frac + zero_frac)] = 0.0
    X[(m >= nan_frac + zero_frac) & (m < nan_frac + zero_frac + 0.01)] = 1e-36
    return X


def quantiles_of(X):
    out = []
    for j in range(X.shape[1]):
        col = X[:, j][~np.isnan(X[:, j])]
        out.append(np.quantile(col, np.linspace(0.02, 0.98, 49)) if len(col) else np.array([0.0, 0.5]))
    return out


def assert_same(got, exp):
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |d| = {np.nanmax(np.abs(got - exp))}"
    assert np.array_equal(np.argsort(-got, kind="stable"), np.argsort(-exp, kind="stable"))


@pytest.mark.parametrize("rows", [1, 63, 64, 100, 257, 1000, 4099])
def test_lightgbm_ranklens_shape(ctx, rows):
    rng = np.random.default_rng(rows)
    X = make_X(rng, rows, 24, cat_col=7)
    blob = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=quantiles_of(X), cat_features=[7], cat_prob=0.1)
    exp = OracleForest.from_lightgbm_text(blob).predict(X)
    b = M.HipBooster(blob, M.LIGHTGBM, ctx)
    info = b.info()
    assert info["n_trees"] == 500 and info["is_f64"] == 1 and info["n_categorical"] > 0
    assert_same(b.predictMat(X.reshape(-1), rows, 24), exp)
    b.close()
    assert b.isClosed()


@pytest.mark.parametrize("fmt", ["json", "ubj"])
@pytest.mark.parametrize("n_trees,depth,cols,rows", [(100, 6, 24, 100), (500, 6, 24, 777), (50, 8, 17, 300)])
def test_xgboost(ctx, fmt, n_trees, depth, cols, rows):
    rng = np.random.default_rng(n_trees + rows)
    X = make_X(rng, rows, cols, cat_col=3)
    blob = synth.synthetic_xgb_model(n_trees=n_trees, n_features=cols, depth=depth, quantiles=quantiles_of(X),
                                     cat_features=[3], cat_prob=0.1, fmt=fmt, complete=(depth <= 6))
    exp = OracleForest.from_xgboost(blob).predict(X)
    b = M.HipBooster(blob, M.XGBOOST, ctx)
    assert b.info()["is_f64"] == 0 and b.info()["base_score"] == 0.5
    assert_same(b.predict(X), exp)


def test_lightgbm_c3_shape_64_columns(ctx):
    rng = np.random.default_rng(3)
    X = make_X(rng, 1000, 64)
    blob = synth.synthetic_lgbm_model(n_trees=500, n_features=64, quantiles=quantiles_of(X))
    assert_same(M.HipBooster(blob, M.LIGHTGBM, ctx).predict(X), OracleForest.from_lightgbm_text(blob).predict(X))


def test_wide_matrix_uses_global_rows_path(ctx):
    rng = np.random.default_rng(4)
    X = make_X(rng, 300, 400)
    blob = synth.synthetic_lgbm_model(n_trees=40, n_features=400, quantiles=quantiles_of(X))
    assert_same(M.HipBooster(blob, M.LIGHTGBM, ctx).predict(X), OracleForest.from_lightgbm_text(blob).predict(X))
    xb = synth.synthetic_xgb_model(n_trees=40, n_features=400, depth=5, quantiles=quantiles_of(X))
    assert_same(M.HipBooster(xb, M.XGBOOST, ctx).predict(X), OracleForest.from_xgboost(xb).predict(X))


def test_deep_large_trees(ctx):
    rng = np.random.default_rng(5)
    X = make_X(rng, 500, 12)
    blob = synth.synthetic_lgbm_model(n_trees=30, n_features=12, num_leaves=255, max_depth=20, quantiles=quantiles_of(X))
    assert_same(M.HipBooster(blob, M.LIGHTGBM, ctx).predict(X), OracleForest.from_lightgbm_text(blob).predict(X))


def test_sklearn_golden(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "sklearn_forest.npz"))
    lg = open(os.path.join(golden_dir, "sklearn_forest.lgbm.txt"), "rb").read()
    xg = open(os.path.join(golden_dir, "sklearn_forest.xgb.json"), "rb").read()
    assert np.array_equal(M.HipBooster(lg, M.LIGHTGBM, ctx).predict(g["X"]), g["expected_f64"])
    assert np.array_equal(M.HipBooster(xg, M.XGBOOST, ctx).predict(g["X"]), g["expected_xgb_f32"])


def test_container_and_feature_mismatch(ctx):
    inner = synth.synthetic_lgbm_model(n_trees=5, n_features=3, seed=9)
    blob = synth.write_container(["a", "b", "c"], 0, inner)
    b = M.HipBooster.from_container(blob, ["a", "b", "c"], ctx)
    X = np.random.default_rng(0).normal(size=(10, 3))
    assert_same(b.predict(X), OracleForest.from_container(blob).predict(X))
    with pytest.raises(M.MrkError) as e:
        M.HipBooster.from_container(blob, ["a", "c", "b"], ctx)
    assert e.value.status == -8 and "booster trained with" in e.value.message
    with pytest.raises(M.MrkError):
        M.HipBooster.from_container(b"\x07" + blob[1:], None, ctx)  # bad bitstream version


def test_error_behaviour(ctx):
    xb = synth.synthetic_xgb_model(n_trees=3, n_features=4, depth=3)
    b = M.HipBooster(xb, M.XGBOOST, ctx)
    X = np.zeros((2, 4))
    X[1, 2] = math.inf
    with pytest.raises(M.MrkError) as e:  # XGBoost rejects inf when missing = NaN
        b.predict(X)
    assert "inf" in e.value.message
    with pytest.raises(M.MrkError) as e:  # fewer columns than the booster splits on
        b.predict(np.zeros((2, 1)))
    assert e.value.status == -4
    with pytest.raises(M.MrkError):
        M.HipBooster(b"not a model", M.LIGHTGBM, ctx)
    with pytest.raises(M.MrkError):
        M.HipBooster(b"binf\x00\x00", M.XGBOOST, ctx)
    b.close()
    with pytest.raises(M.MrkError):
        b.predict(np.zeros((1, 4)))
    assert b.predictMat.__self__.isClosed()


def test_full_size_properties_100k(ctx):
    """C4-sized batch (100 000 candidates): properties that do not need the oracle at full size."""
    rng = np.random.default_rng(6)
    X = make_X(rng, 100_000, 24, cat_col=7)
    blob = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=quantiles_of(X), cat_features=[7], cat_prob=0.1)
    b = M.HipBooster(blob, M.LIGHTGBM, ctx)
    full = b.predict(X)
    # (1) a row's score does not depend on its position or batch: permutation equivariance
    perm = rng.permutation(len(X))
    assert np.array_equal(b.predict(X[perm]), full[perm])
    # (2) tiling independence: two halves == the whole
    assert np.array_equal(np.concatenate([b.predict(X[:33_333]), b.predict(X[33_333:])]), full)
    # (3) the oracle on a seeded sample
    idx = rng.choice(len(X), 2000, replace=False)
    assert np.array_equal(full[idx], OracleForest.from_lightgbm_text(blob).predict(X[idx]))
