"""CPU tests of the forest oracle (oracle/forest_oracle.cpp + oracle/formats.py).

The reference never asserts a forest score — these pins are (i) hand-computed
known answers for every LightGBM / XGBoost decision rule and (ii) scikit-learn as an independent
evaluator on forests exported into both on-disk formats (tests/golden/make_sklearn_golden.py).
"""
import math
import os

import numpy as np
import pytest

from workloads import synth
from oracle.forest import InfInData, OracleForest

NAN = float("nan")


def lgbm_one_tree(**kw):
    t = dict(num_leaves=2, split_feature=[0], threshold=[0.5], decision_type=[0], left_child=[~0], right_child=[~1],
             leaf_value=[1.0, 2.0])
    t.update(kw)
    return OracleForest.from_lightgbm_text(synth.write_lightgbm_text([t], 2))


def test_lgbm_numerical_le_goes_left():
    f = lgbm_one_tree()
    X = np.array([[0.5, 0], [0.5000001, 0], [-3, 0]], dtype=np.float64)
    assert f.predict(X).tolist() == [1.0, 2.0, 1.0]  # fval <= threshold -> left


@pytest.mark.parametrize("missing_type,default_left,x,expected", [
    # MissingType::None (0): NaN is replaced by 0.0 and compared with the threshold
    (0, False, NAN, 1.0), (0, True, NAN, 1.0),
    # MissingType::Zero (1): zero (|x| <= 1e-35) and NaN (-> 0.0 -> IsZero) take the default direction
    (1, False, 0.0, 2.0), (1, True, 0.0, 1.0), (1, False, NAN, 2.0), (1, True, NAN, 1.0), (1, False, 1e-36, 2.0),
    (1, False, 0.25, 1.0), (1, True, 0.75, 2.0),
    # MissingType::NaN (2): only NaN takes the default direction; 0.0 is compared
    (2, False, NAN, 2.0), (2, True, NAN, 1.0), (2, False, 0.0, 1.0), (2, True, 0.75, 2.0),
])
def test_lgbm_missing_types(missing_type, default_left, x, expected):
    dt = (missing_type << 2) | (2 if default_left else 0)
    f = lgbm_one_tree(decision_type=[dt])
    assert f.predict(np.array([[x, 0.0]])).tolist() == [expected]


def test_lgbm_none_missing_nan_with_negative_threshold():
    f = lgbm_one_tree(threshold=[-0.5])  # NaN -> 0.0 ; 0.0 <= -0.5 is false -> right
    assert f.predict(np.array([[NAN, 0.0]])).tolist() == [2.0]


def test_lgbm_dense_row_zero_flush():
    # c_api RowFunctionFromDenseMatric drops |x| <= 1e-35: 1e-36 reads back as 0.0, so with a
    # threshold of 1e-37 the row goes LEFT (0.0 <= 1e-37) although 1e-36 > 1e-37.
    f = lgbm_one_tree(threshold=[1e-37])
    assert f.predict(np.array([[1e-36, 0.0], [1e-30, 0.0]])).tolist() == [1.0, 2.0]


def test_lgbm_categorical():
    # categories {1, 5, 33} -> left ; bitset words: word0 = 1<<1 | 1<<5, word1 = 1<<1
    t = dict(num_leaves=2, split_feature=[1], threshold=[0.0], decision_type=[1], left_child=[~0], right_child=[~1],
             leaf_value=[10.0, 20.0], cat_boundaries=[0, 2], cat_threshold=[(1 << 1) | (1 << 5), 1 << 1])
    f = OracleForest.from_lightgbm_text(synth.write_lightgbm_text([t], 2))
    xs = [1.0, 5.0, 33.0, 5.9, 0.0, 2.0, 64.0, -1.0, NAN, 1e300, -1e300]
    exp = [10.0, 10.0, 10.0, 10.0, 20.0, 20.0, 20.0, 20.0, 20.0, 20.0, 20.0]
    X = np.array([[0.0, x] for x in xs])
    assert f.predict(X).tolist() == exp


def test_lgbm_single_leaf_tree_and_sum_order():
    t1 = dict(num_leaves=1, leaf_value=[0.1])
    t2 = dict(num_leaves=1, leaf_value=[0.2])
    t3 = dict(num_leaves=1, leaf_value=[0.3])
    f = OracleForest.from_lightgbm_text(synth.write_lightgbm_text([t1, t2, t3], 1))
    assert f.predict(np.zeros((1, 1)))[0] == (0.0 + 0.1 + 0.2) + 0.3  # f64, tree order


def xgb_one_tree(**kw):
    t = dict(left_children=[1, -1, -1], right_children=[2, -1, -1], split_indices=[0, 0, 0],
             split_conditions=[0.5, 1.0, 2.0], default_left=[0, 0, 0], split_type=[0, 0, 0], categories=[],
             categories_nodes=[], categories_segments=[], categories_sizes=[], parents=[2147483647, 0, 0],
             base_weights=[0.0] * 3, loss_changes=[0.0] * 3, sum_hessian=[1.0] * 3, id=0,
             tree_param={"num_deleted": "0", "num_feature": "2", "num_nodes": "3", "size_leaf_vector": "1"})
    t.update(kw)
    return t


def test_xgb_numerical_lt_goes_left_and_base_score():
    f = OracleForest.from_xgboost(synth.write_xgboost_json(synth.xgboost_document([xgb_one_tree()], 2, 0.5)))
    X = np.array([[0.5, 0], [0.49999997, 0], [NAN, 0]])
    # fvalue < split_condition -> left ; 0.5 is NOT < 0.5 ; NaN is missing -> default (right)
    assert f.predict(X).tolist() == [2.5, 1.5, 2.5]
    f2 = OracleForest.from_xgboost(synth.write_xgboost_json(
        synth.xgboost_document([xgb_one_tree(default_left=[1, 0, 0])], 2, 0.5)))
    assert f2.predict(np.array([[NAN, 0]])).tolist() == [1.5]


def test_xgb_features_are_narrowed_to_f32():
    # 0.5 - 1e-12 rounds to 0.5f, which is not < 0.5f
    f = OracleForest.from_xgboost(synth.write_xgboost_json(synth.xgboost_document([xgb_one_tree()], 2, 0.5)))
    assert f.predict(np.array([[0.5 - 1e-12, 0]])).tolist() == [2.5]


def test_xgb_f32_accumulation():
    trees = [xgb_one_tree(split_conditions=[0.5, 0.1, 0.1]) for _ in range(10)]
    f = OracleForest.from_xgboost(synth.write_xgboost_json(synth.xgboost_document(trees, 2, 0.5)))
    acc = np.float32(0.5)
    for _ in range(10):
        acc = np.float32(acc + np.float32(0.1))
    assert f.predict(np.zeros((1, 2)))[0] == float(acc)
    assert f.predict(np.zeros((1, 2)))[0] != 0.5 + 10 * 0.1


def test_xgb_categorical_members_go_right():
    t = xgb_one_tree(split_type=[1, 0, 0], split_indices=[1, 0, 0], categories=[2, 7], categories_nodes=[0],
                     categories_segments=[0], categories_sizes=[2], split_conditions=[0.0, 1.0, 2.0])
    f = OracleForest.from_xgboost(synth.write_xgboost_json(synth.xgboost_document([t], 2, 0.0)))
    xs = [2.0, 7.0, 7.5, 0.0, 3.0, 100.0, -1.0, 16777216.0, NAN]
    exp = [2.0, 2.0, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 2.0]  # NaN -> default (right, default_left=0)
    assert f.predict(np.array([[0.0, x] for x in xs])).tolist() == exp


def test_xgb_inf_is_rejected():
    f = OracleForest.from_xgboost(synth.write_xgboost_json(synth.xgboost_document([xgb_one_tree()], 2, 0.5)))
    with pytest.raises(InfInData):
        f.predict(np.array([[math.inf, 0.0]]))
    with pytest.raises(InfInData):
        f.predict(np.array([[1e300, 0.0]]))  # overflows the f32 narrowing


def test_xgb_json_and_ubjson_agree():
    q = None
    js = synth.synthetic_xgb_model(n_trees=7, n_features=5, depth=4, fmt="json", cat_features=[2], cat_prob=0.3, seed=3)
    ub = synth.synthetic_xgb_model(n_trees=7, n_features=5, depth=4, fmt="ubj", cat_features=[2], cat_prob=0.3, seed=3)
    rng = np.random.default_rng(1)
    X = rng.normal(size=(200, 5))
    X[:, 2] = rng.integers(0, 16, size=200)
    X[rng.random(X.shape) < 0.05] = NAN
    a = OracleForest.from_xgboost(js).predict(X)
    b = OracleForest.from_xgboost(ub).predict(X)
    assert np.array_equal(a, b)


def test_xgb_legacy_binary_agrees_with_json():
    """The legacy binary serialisation (xgboost4j < 2.0's Booster.toByteArray()) of a model predicts like its JSON form:
    the writer (workloads/synth.py) and the oracle's reader (oracle/formats.py) restate learner.cc / gbtree_model.cc /
    tree_model.cc independently of the product's reader (csrc/forest.cpp), with and without the "binf" header and the
    pre-1.0 leaf vector."""
    rng = np.random.default_rng(2)
    trees = [synth.random_xgb_tree(np.random.Generator(np.random.PCG64(k)), 5, 4, None, None, 0.0, 16, k % 2 == 0) for k in range(9)]
    doc = synth.xgboost_document(trees, 5, 0.5)
    X = rng.normal(size=(300, 5))
    X[rng.random(X.shape) < 0.05] = NAN
    want = OracleForest.from_xgboost(synth.write_xgboost_json(doc)).predict(X)
    for binf in (False, True):
        for lv in (False, True):
            blob = synth.write_xgboost_legacy(doc, binf=binf, leaf_vector=lv)
            assert blob[:1] != b"{"
            assert np.array_equal(OracleForest.from_xgboost(blob).predict(X), want), (binf, lv)


def test_container_roundtrip():
    inner = synth.synthetic_lgbm_model(n_trees=3, n_features=4, seed=5)
    blob = synth.write_container(["a", "b", "c"], 0, inner, version=3)
    f = OracleForest.from_container(blob)
    assert f.container_features == ["a", "b", "c"] and f.n_trees == 3
    blob2 = synth.write_container(["a"], 1, synth.synthetic_xgb_model(n_trees=2, n_features=4, depth=2), version=2)
    assert OracleForest.from_container(blob2).n_trees == 2


def test_sklearn_golden_lightgbm_format(golden_dir):
    g = np.load(os.path.join(golden_dir, "sklearn_forest.npz"))
    f = OracleForest.from_lightgbm_text(open(os.path.join(golden_dir, "sklearn_forest.lgbm.txt"), "rb").read())
    got = f.predict(g["X"])
    assert np.array_equal(got, g["expected_f64"])  # bit-exact: same f64 additions in tree order


def test_sklearn_golden_xgboost_format(golden_dir):
    g = np.load(os.path.join(golden_dir, "sklearn_forest.npz"))
    f = OracleForest.from_xgboost(open(os.path.join(golden_dir, "sklearn_forest.xgb.json"), "rb").read())
    got = f.predict(g["X"])
    assert np.array_equal(got, g["expected_xgb_f32"])  # leaves chosen by sklearn.apply, f32 sum from base 0.5
    assert np.allclose(got - 0.5, g["expected_f64"], atol=1e-5)
