"""GPU parity: the HIP forest scorer (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): identical ordering, |score delta| <= 1e-5; the kernel adds leaves in
tree order in the library's own precision, so these tests demand bit-exact scores.
"""
import math
import os

import numpy as np
import pytest

import metarank_amd as M
from workloads import synth
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


@pytest.mark.parametrize("binf", [False, True])
def test_xgboost_legacy_binary(ctx, binf):
    """XGBoostBooster(bytes) of a model stored by xgboost4j < 2.0 (LambdaMARTRanker.scala:229-230 hands the booster
    whatever toByteArray() wrote): the legacy binary serialisation loads and scores like the same model as JSON."""
    rng = np.random.default_rng(12)
    X = make_X(rng, 500, 20)
    trees = [synth.random_xgb_tree(np.random.Generator(np.random.PCG64(k)), 20, 6, quantiles_of(X), None, 0.0, 16, k % 3 != 0) for k in range(60)]
    doc = synth.xgboost_document(trees, 20, 0.5)
    js = synth.write_xgboost_json(doc)
    blob = synth.write_xgboost_legacy(doc, binf=binf, leaf_vector=binf)
    exp = OracleForest.from_xgboost(js).predict(X)
    b = M.HipBooster(blob, M.XGBOOST, ctx)
    assert b.info()["is_f64"] == 0 and b.info()["base_score"] == 0.5 and b.info()["n_trees"] == 60
    assert_same(b.predict(X), exp)
    assert_same(M.HipBooster(js, M.XGBOOST, ctx).predict(X), exp)


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


# ---------------------------------------------------------------------------------------------
# The two scorers: bit-vector (trees of <= 16 leaves, score_qs.hip) and tree-walk (score.hip).
# Both must reproduce the oracle bit for bit on the same model; the library reads MRK_SCORER / MRK_QS_KERNEL once, so a
# test that pins a kernel re-reads them (M.reload_switches).
@pytest.fixture
def scorer_env():
    saved = {k: os.environ.get(k) for k in ("MRK_SCORER", "MRK_QS_KERNEL", "MRK_QS_R", "MRK_QS_SPLIT", "MRK_WALK_TILE")}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    M.reload_switches()


def _predict_with(b, X, **env):
    for k in ("MRK_SCORER", "MRK_QS_KERNEL", "MRK_QS_R", "MRK_QS_SPLIT", "MRK_WALK_TILE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    M.reload_switches()
    return b.predict(X)


def _all_kernels(b, X):
    out = {}
    for nw in ("1", "2", "4", "8", "16"):  # 1 = one wavefront per tile; 2/4/8/16 = wavefronts splitting the trees of a tile
        out[f"bitvector-wave-split{nw}"] = _predict_with(b, X, MRK_QS_KERNEL="1", MRK_QS_SPLIT=nw)
    out["bitvector-wave-auto"] = _predict_with(b, X, MRK_QS_KERNEL="1")
    for r in ("2", "4", "8"):
        out[f"bitvector-generic-r{r}"] = _predict_with(b, X, MRK_QS_KERNEL="0", MRK_QS_R=r)
    out["walk"] = _predict_with(b, X, MRK_SCORER="walk")
    out["walk-256-row-tiles"] = _predict_with(b, X, MRK_SCORER="walk", MRK_WALK_TILE="256")
    _predict_with(b, X[:1])
    return out


@pytest.mark.parametrize("rows", [1, 2, 127, 128, 129, 1000])
@pytest.mark.parametrize("num_leaves,missing,cat_prob", [(16, "per_node", 0.02), (16, "per_feature", 0.0), (7, "per_node", 0.05),
                                                         (2, "per_feature", 0.3), (1, "per_node", 0.0)])
def test_bitvector_and_walk_kernels_lightgbm(ctx, scorer_env, rows, num_leaves, missing, cat_prob):
    rng = np.random.default_rng(rows * 31 + num_leaves)
    X = make_X(rng, rows, 24, cat_col=7, n_cats=40)
    blob = synth.synthetic_lgbm_model(n_trees=120, n_features=24, num_leaves=num_leaves, quantiles=quantiles_of(X),
                                      cat_features=[7], cat_prob=cat_prob, n_cats=40, missing=missing, seed=rows + num_leaves)
    exp = OracleForest.from_lightgbm_text(blob).predict(X)
    b = M.HipBooster(blob, M.LIGHTGBM, ctx)
    assert b.info()["bitvector"] == 1 and (b.info()["tile_columns"] > 0) == (num_leaves > 1)
    for name, got in _all_kernels(b, X).items():
        assert np.array_equal(got, exp), name


@pytest.mark.parametrize("rows", [1, 128, 333])
@pytest.mark.parametrize("depth,complete", [(4, True), (4, False), (2, True)])
def test_bitvector_and_walk_kernels_xgboost(ctx, scorer_env, rows, depth, complete):
    rng = np.random.default_rng(rows + depth)
    X = make_X(rng, rows, 17, cat_col=3)
    blob = synth.synthetic_xgb_model(n_trees=90, n_features=17, depth=depth, quantiles=quantiles_of(X), cat_features=[3],
                                     cat_prob=0.1, complete=complete, seed=rows)
    exp = OracleForest.from_xgboost(blob).predict(X)
    b = M.HipBooster(blob, M.XGBOOST, ctx)
    assert b.info()["bitvector"] == 1
    for name, got in _all_kernels(b, X).items():
        assert np.array_equal(got, exp), name


def test_large_trees_have_no_bitvector_image(ctx):
    blob = synth.synthetic_xgb_model(n_trees=5, n_features=6, depth=6)
    assert M.HipBooster(blob, M.XGBOOST, ctx).info()["bitvector"] == 0
    blob = synth.synthetic_lgbm_model(n_trees=5, n_features=6, num_leaves=17, max_depth=10)
    assert M.HipBooster(blob, M.LIGHTGBM, ctx).info()["bitvector"] == 0


def test_bitvector_special_values(ctx, scorer_env):
    """-0.0, +-1e-36 (LightGBM's zero flush), +-inf, huge categories, thresholds hit exactly."""
    rng = np.random.default_rng(77)
    X = make_X(rng, 600, 8, cat_col=2, n_cats=70)
    q = quantiles_of(X)
    specials = [0.0, -0.0, 1e-36, -1e-36, 1e-35, math.inf, -math.inf, NAN, 1e300, -1e300, 2.0 ** 31, -(2.0 ** 31), 16777216.0, 70000.0]
    for i, v in enumerate(specials):
        X[i, :] = v
    for j in range(8):  # values equal to thresholds
        X[100:100 + len(q[j]), j] = q[j]
    blob = synth.synthetic_lgbm_model(n_trees=200, n_features=8, quantiles=q, cat_features=[2], cat_prob=0.1, n_cats=70, seed=5)
    exp = OracleForest.from_lightgbm_text(blob).predict(X)
    b = M.HipBooster(blob, M.LIGHTGBM, ctx)
    for name, got in _all_kernels(b, X).items():
        assert np.array_equal(got, exp), name
    Xf = X.copy()
    Xf[np.isinf(Xf)] = 3.0e38  # XGBoost rejects inf; keep a value that narrows to a finite float
    Xf[np.abs(Xf) > 3.0e38] = 3.0e38
    xb = synth.synthetic_xgb_model(n_trees=150, n_features=8, depth=4, quantiles=q, cat_features=[2], cat_prob=0.1, n_cats=70, seed=6)
    expx = OracleForest.from_xgboost(xb).predict(Xf)
    bx = M.HipBooster(xb, M.XGBOOST, ctx)
    for name, got in _all_kernels(bx, Xf).items():
        assert np.array_equal(got, expx), name
    Xi = Xf.copy()
    Xi[5, 1] = math.inf
    with pytest.raises(M.MrkError) as e:
        bx.predict(Xi)
    assert "inf" in e.value.message


def test_xgboost_rejects_inf_in_a_column_the_forest_never_splits_on(ctx, scorer_env):
    """XGBoost's DMatrix rejects a row for an inf in ANY column, so every scorer path must too - also for a column no
    tree uses and for one beyond the model's num_feature (bit-vector binning, tree walk with the LDS tile, tree walk over
    a matrix too wide for it)."""
    trees = [synth.random_xgb_tree(np.random.Generator(np.random.PCG64(k)), 3, 3, None, None, 0.0, 16, True) for k in range(5)]
    for shallow in (True, False):  # <= 16 leaves: bit-vector image; deeper: tree walk only
        if not shallow:
            trees = [synth.random_xgb_tree(np.random.Generator(np.random.PCG64(k)), 3, 6, None, None, 0.0, 16, True) for k in range(5)]
        for cols in (6, 3000):     # 3000 columns: the rows do not fit the walk's LDS tile
            blob = synth.write_xgboost_json(synth.xgboost_document(trees, 3, 0.5))   # splits on columns 0..2 only
            b = M.HipBooster(blob, M.XGBOOST, ctx)
            X = np.zeros((5, cols))
            assert np.isfinite(b.predict(X)).all()
            for col in (4, cols - 1):
                Xi = X.copy()
                Xi[3, col] = -math.inf
                for env in ({}, {"MRK_SCORER": "walk"}):
                    with pytest.raises(M.MrkError) as e:
                        _predict_with(b, Xi, **env)
                    assert "inf" in e.value.message, (shallow, cols, col, env)
            _predict_with(b, X)
            b.close()
