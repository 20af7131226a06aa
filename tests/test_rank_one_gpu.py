"""mrk_rank's ONE-LAUNCH path (csrc/rank_device.hpp rank_one_body: pre-pass + assembly + forest + ordering of a small
request in the request's workgroup, results straight into pinned memory) against the oracle and against the three-launch
path it replaces (MRK_RANK_ONE=0): same scores, same order, same per-request errors - specialised and interpreting
kernel, LightGBM f64 and XGBoost f32 forests, categorical splits, requests of 0 / 1 / 128 / 129 candidates."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import metarank_amd as M
from backends import HipBackend, OracleBackend
from workloads import ranklens, synth

N_ITEMS, N_SESS = 3000, 300


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def with_env(env: dict):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    M.reload_switches()
    return saved


def restore_env(saved: dict):
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    M.reload_switches()


def requests():
    reqs = ranklens.generate_requests(6, 100, N_ITEMS, N_SESS, seed=81)
    reqs += ranklens.generate_requests(1, 1, N_ITEMS, N_SESS, seed=82) + ranklens.generate_requests(1, 128, N_ITEMS, N_SESS, seed=83)
    reqs += ranklens.generate_requests(1, 129, N_ITEMS, N_SESS, seed=84)   # one candidate too many: the three-launch path
    reqs.append({"id": "none", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": []})
    reqs.append({"id": "odd", "timestamp": ranklens.TS, "user": None, "session": reqs[0]["session"], "fields": [],
                 "items": [{"id": "nobody"}, {"id": "12"}, {"id": "12"}, {"id": ""}]})
    return reqs


@pytest.mark.gpu
@pytest.mark.parametrize("jit", ["1", "0"])
@pytest.mark.parametrize("kind", ["lgbm", "xgb4"])
def test_one_launch_equals_three_launches_and_the_oracle(kind, jit):
    saved = with_env({"MRK_RANK_JIT": jit})
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
        reqs = requests()
        q = ranklens.column_quantiles(np.concatenate([orc.matrix(ev) for ev in reqs[:6]]))
        if kind == "lgbm":
            blob, backend = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.05, missing="per_feature"), 0
        else:
            blob, backend = synth.synthetic_xgb_model(n_trees=137, n_features=24, depth=4, quantiles=q, cat_features=[7], cat_prob=0.05), 1
        orc.load_model(blob, backend)
        hip.load_model(blob, backend)
        assert hip.booster.info()["bitvector"] == 1
        got = {}
        for one in ("1", "0"):
            s2 = with_env({"MRK_RANK_ONE": one})
            got[one] = [hip.ranker.rerank("xgboost", ev, hip.booster) for ev in reqs]
            restore_env(s2)
        for k, ev in enumerate(reqs):
            _, es, eo = orc.rerank(ev)
            for one in ("1", "0"):
                _, s, o = got[one][k]
                assert same(s, es) and o.tolist() == eo.tolist(), (kind, jit, one, k)
        # concurrent callers: the batching front hands the one-launch kernel up to 16 requests at a time
        with ThreadPoolExecutor(12) as ex:
            res = list(ex.map(lambda ev: hip.ranker.rerank("xgboost", ev, hip.booster), reqs * 3))
        for k, (_, s, o) in enumerate(res):
            _, es, eo = got["0"][k % len(reqs)]
            assert same(s, es) and o.tolist() == eo.tolist(), k
        # a request the reference throws on: the same error from both paths (normalised rate: global clicks == 0)
        hip.put_periodic("global/ctr_click_norm", [0, 5])
        for one in ("1", "0"):
            s2 = with_env({"MRK_RANK_ONE": one})
            with pytest.raises(M.MrkError) as ei:
                hip.ranker.rerank("xgboost", reqs[0], hip.booster)
            assert ei.value.status == -5
            restore_env(s2)
    finally:
        restore_env(saved)
        hip.close()


@pytest.mark.gpu
def test_one_launch_reports_an_xgboost_inf_like_the_batch_path():
    cfg = ranklens.ranklens_config()
    hip = HipBackend(cfg, "xgboost")
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        ev = ranklens.generate_requests(1, 50, N_ITEMS, N_SESS, seed=85)[0]
        blob = synth.synthetic_xgb_model(n_trees=20, n_features=24, depth=3)
        hip.load_model(blob, 1)
        hip.put_double(f"item={ev['items'][3]['id']}/popularity", 1e300)   # +inf after the Double -> Float narrowing
        for one in ("1", "0"):
            saved = with_env({"MRK_RANK_ONE": one})
            with pytest.raises(M.MrkError) as ei:
                hip.ranker.rerank("xgboost", ev, hip.booster)
            assert ei.value.status == -1 and "inf" in ei.value.message
            restore_env(saved)
    finally:
        hip.close()
