"""The multi-workgroup ordering of requests with more than 4 096 candidates (csrc/bigsort.hip: sample sort on (key,
index) pairs) against numpy's stable argsort on the java.lang.Double.compare key of the negated score
(ml/Ranker.scala:52-67: `sortBy(-_.score)`): every distribution that could unbalance the buckets, every size class (one
level, a sample that needs the sort itself, several requests in one batch), the global-memory path of an outgrown
bucket - and `norm: position` over a request larger than one workgroup sorts (ml/onnx/Normalize.scala:25-40)."""
import ctypes as C
import os

import numpy as np
import pytest

import metarank_amd as M
from backends import HipBackend, OracleBackend
from workloads import ranklens, synth


def expected_order(scores: np.ndarray) -> np.ndarray:
    k = -scores
    bits = k.view(np.uint64).copy()
    bits[np.isnan(k)] = 0x7FF8000000000000
    neg = (bits >> np.uint64(63)) != 0
    key = np.where(neg, ~bits, bits | np.uint64(1 << 63))
    return np.argsort(key, kind="stable").astype(np.int32)


def distributions(n: int, rng):
    yield "normal", rng.normal(size=n)
    yield "all equal", np.zeros(n)
    yield "three values", rng.integers(0, 3, size=n).astype(np.float64)
    yield "ascending", np.arange(n, dtype=np.float64)
    yield "descending", -np.arange(n, dtype=np.float64)
    x = rng.normal(size=n)
    x[rng.random(n) < 0.01] = np.nan
    x[rng.random(n) < 0.01] = np.inf
    x[rng.random(n) < 0.01] = -np.inf
    x[rng.random(n) < 0.02] = 0.0
    x[rng.random(n) < 0.02] = -0.0
    yield "specials", x
    yield "ties in the first half", np.concatenate([np.full(n // 2, 1.5), rng.normal(size=n - n // 2)])
    yield "saw-tooth", (np.arange(n) % 97).astype(np.float64) * 1e-3
    yield "wide exponents", rng.normal(size=n) * np.exp2(rng.integers(-300, 300, size=n).astype(np.float64))
    y = rng.normal(size=n)
    y[:: max(n // 8192, 1)] = 1e9   # the strata's first candidates are outliers: what a strided sample would have drawn
    yield "outliers on a stride", y


class DeviceScores:
    """a batch of unknown items whose device score buffer the test overwrites before mrk_batch_sort"""

    def __init__(self, hip, sizes):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        reqs = [{"id": f"r{k}", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [],
                 "items": [{"id": f"x{i}"} for i in range(n)]} for k, n in enumerate(sizes)]
        self.batch = hip.ranker.new_batch()
        self.batch.load("xgboost", M.RequestSet(reqs, pinned=False))
        self.batch.run(None)   # NoopModel: every score 0.0 - the all-ties case sorts here already
        self.batch.sync()
        self.d_scores = self.batch.device_outputs()[0]

    def order_of(self, scores: np.ndarray) -> np.ndarray:
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        assert self.hip.hipMemcpy(self.d_scores, scores.ctypes.data, scores.nbytes, 1) == 0
        self.batch.sort()
        return self.batch.fetch()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4097, 5000, 100_000, 600_000])
def test_large_requests_are_ordered_like_a_stable_sort(n):
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        dev = DeviceScores(hip, [n])
        assert dev.batch.fetch()[1].tolist() == list(range(n))   # all scores 0.0: request order
        rng = np.random.default_rng(n)
        for name, s in distributions(n, rng):
            got = dev.order_of(s)
            assert np.array_equal(got, expected_order(s)), (n, name)
        dev.batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_several_large_and_small_requests_in_one_batch():
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        sizes = [100, 30_000, 4096, 4097, 0, 9000, 1]
        dev = DeviceScores(hip, sizes)
        rng = np.random.default_rng(5)
        s = rng.normal(size=sum(sizes)).round(2)   # two decimals: plenty of ties
        got = dev.order_of(s)
        off = 0
        for n in sizes:
            assert np.array_equal(got[off:off + n], expected_order(s[off:off + n])), n
            off += n
        dev.batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"MRK_BIG_SORT_BUCKET": "256"}, {"MRK_BIG_SORT_TILE": "256"}, {"MRK_BIG_SORT_BUCKET": "128", "MRK_BIG_SORT_TILE": "2048"}])
def test_every_launch_shape_of_the_sample_sort_gives_the_same_order(env):
    """More / smaller workgroups per classify pass, smaller buckets (more splitters, shorter local sorts): the order is the stable
    sort's whatever the shape."""
    for k, v in env.items():
        os.environ[k] = v
    M.reload_switches()
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        for n in (100_000, 7000):
            dev = DeviceScores(hip, [n])
            rng = np.random.default_rng(n + 1)
            for name, s in distributions(n, rng):
                assert np.array_equal(dev.order_of(s), expected_order(s)), (env, n, name)
            dev.batch.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
        M.reload_switches()
        hip.close()


@pytest.mark.gpu
def test_a_bucket_that_outgrows_lds_is_sorted_in_global_memory():
    os.environ["MRK_BIG_SORT_CAP"] = "300"   # buckets hold ~1 000 pairs: every one takes the global-memory path
    M.reload_switches()
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        n = 20_000
        dev = DeviceScores(hip, [n])
        rng = np.random.default_rng(9)
        for name, s in distributions(n, rng):
            assert np.array_equal(dev.order_of(s), expected_order(s)), name
        dev.batch.close()
    finally:
        os.environ.pop("MRK_BIG_SORT_CAP", None)
        M.reload_switches()
        hip.close()


@pytest.mark.gpu
def test_norm_position_over_a_request_larger_than_one_workgroup_sorts():
    """round 2 answered MRK_ERR_UNSUPPORTED here; the reference has no such limit (Normalize.scala:25-40)"""
    cfg = ranklens.c5_config()
    cfg["features"][-1]["norm"] = "position"
    n_items = 1500
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(n_items, 100))
            ranklens.load_state(b, ranklens.c5_embeddings(n_items))
        reqs = ranklens.generate_requests(1, 5000, n_items, 100, seed=61) + ranklens.generate_requests(2, 60, n_items, 100, seed=62)
        for k, ev in enumerate(reqs):
            ev["fields"] = [{"name": "__embedding:title_match", "value": ranklens.c5_query(seed=k)}]
        reqs[0]["items"][17] = {"id": "nobody"}   # no embedding: NaN stays NaN, and still takes a place in the order
        mats = [orc.matrix(ev) for ev in reqs]
        blob = synth.synthetic_lgbm_model(n_trees=100, n_features=25, quantiles=ranklens.column_quantiles(np.concatenate(mats)))
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, mat = batch.fetch(matrix=True)
        assert (batch.status() == 0).all()
        for r, ev in enumerate(reqs):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            _, es, eo = orc.rerank(ev)
            a, b = mat[lo:hi], mats[r]
            assert a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all()), r
            assert np.array_equal(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), r
        # item-sharded 3 ways (round 2: MRK_ERR_UNSUPPORTED for a model with a normalised column): a shard normalises the
        # whole request's column and scores its slice - the bytes of the unsharded run
        b2 = hip.ranker.prepare("xgboost", reqs)
        for rank in range(3):
            b2.run_shard(hip.booster, rank, 3)
        b2.sort()
        s2, o2, m2 = b2.fetch(matrix=True)
        assert (b2.status() == 0).all()
        assert np.array_equal(s2, scores) and np.array_equal(o2, order)
        assert bool(((m2 == mat) | (np.isnan(m2) & np.isnan(mat))).all())
        b2.close()
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_numeric_diversity_over_more_values_than_the_pre_pass_sorts():
    """`diversity` with `top` above 4 096 over a request with that many present numeric values: round 2 answered
    MRK_ERR_UNSUPPORTED; now the host - which holds the same values in its mirror - takes the median (commons-math LEGACY
    percentile, DiversityFeature.scala:113-126) and hands it to the device as a finished pre-pass result."""
    from backends import single_feature_config

    cfg = single_feature_config({"name": "div_pop", "type": "diversity", "source": "item.popularity", "top": 100000})
    orc, hip = OracleBackend(cfg, list(cfg["models"])[0]), HipBackend(cfg, list(cfg["models"])[0])
    try:
        rng = np.random.default_rng(3)
        n = 6000
        vals = rng.normal(size=n).round(3)
        for b in (orc, hip):
            for i in range(n):
                if i % 17 != 5:   # some candidates have no state: NaN in the column, not part of the median
                    b.put_double(f"item=i{i}/div_pop", float(vals[i]) if i % 101 else 0.0)
        items = [{"id": f"i{i}"} for i in range(n)]
        big = {"id": "big", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": items}
        small = {"id": "small", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": items[:50]}
        for ev in (big, small, dict(big, id="5000", items=items[:5000])):
            a, b = hip.matrix(ev), orc.matrix(ev)
            assert a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all()), ev["id"]
            assert np.isnan(b).any() and np.isfinite(b).any()
    finally:
        hip.close()
