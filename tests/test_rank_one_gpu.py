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
        # ... again and again: the kernel's waits are counted (`vmcnt(n)` behind an LDS-DMA), and a wait that is one too weak
        # shows up as a wrong cell once in many launches (round 4 found one that way: cells stored to LDS are not vm operations)
        many = ranklens.generate_requests(60, 100, N_ITEMS, N_SESS, seed=1234)
        want = [orc.rerank(ev) for ev in many]
        for rep in range(3):
            for k, ev in enumerate(many):
                _, s, o = hip.ranker.rerank("xgboost", ev, hip.booster)
                assert same(s, want[k][1]) and o.tolist() == want[k][2].tolist(), (kind, jit, "repeat", rep, k)
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


@pytest.mark.gpu
@pytest.mark.parametrize("jit", ["1", "0"])
def test_serving_queue_equals_mrk_rank(jit):
    """mrk_serve_rank (persistent workgroups polling slots in pinned memory) gives mrk_rank's bytes and errors: requests
    the queue takes, requests it hands to mrk_rank (129 candidates), many host threads over few slots, store puts
    between requests (the workgroups are stopped for the flush and come back), idle workgroups that left."""
    import time

    saved = with_env({"MRK_RANK_JIT": jit, "MRK_SERVE_IDLE_US": "300"})
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    srv = None
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
        reqs = requests()
        q = ranklens.column_quantiles(np.concatenate([orc.matrix(ev) for ev in reqs[:6]]))
        blob = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.05, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        srv = hip.ranker.serve("xgboost", hip.booster, n_slots=3)
        expected = [orc.rerank(ev) for ev in reqs]
        for rep in range(3):
            for k, ev in enumerate(reqs):
                s, o = srv.rerank(ev)
                assert same(s, expected[k][1]) and o.tolist() == expected[k][2].tolist(), (rep, k)
            time.sleep(0.01)   # longer than the idle time: the workgroups leave and are relaunched
        st = srv.stats()
        assert st["fallback"] == 3 and st["queue"] == 3 * (len(reqs) - 1) and st["launches"] >= 3, st   # the 129-candidate request
        with ThreadPoolExecutor(8) as ex:
            res = list(ex.map(srv.rerank, reqs * 6))
        for k, (s, o) in enumerate(res):
            e = expected[k % len(reqs)]
            assert same(s, e[1]) and o.tolist() == e[2].tolist(), k
        # a put between requests: visible to the next request through the queue
        it = reqs[0]["items"][0]["id"]
        for b in (orc, hip):
            b.put_double(f"item={it}/popularity", 987654.0)
            b.put_periodic(f"item={it}/ctr_click", [50, 60])
        _, es, eo = orc.rerank(reqs[0])
        s, o = srv.rerank(reqs[0])
        assert same(s, es) and o.tolist() == eo.tolist()
        # and a request the reference throws on
        for b in (orc, hip):
            b.put_periodic("global/ctr_click_norm", [0, 5])
        with pytest.raises(M.MrkError) as ei:
            srv.rerank(reqs[1])
        assert ei.value.status == -5
    finally:
        if srv is not None:
            srv.close()
        restore_env(saved)
        hip.close()


@pytest.mark.gpu
def test_serving_gangs_leave_and_come_back_as_a_whole():
    """The queue's slots are launched in gangs (one kernel of 8 workgroups per stream).  20 slots = three gangs (8 + 8 + 4); 12
    threads use slots of two of them, pause for longer than the workgroups' life and idle time, and go on: every slot that was left
    is found left by its next request and its WHOLE gang is launched again (launches grow by gangs, not by slots), requests
    published while a gang is being revived are answered by the new launch, and mrk_rank - called while the queue is started -
    is answered through the queue too.  Every result equals the oracle's."""
    import threading
    import time

    saved = with_env({"MRK_RANK_JIT": "1", "MRK_SERVE_IDLE_US": "1500", "MRK_SERVE_LIFE_US": "5000", "MRK_SERVE_SPIN_CALLERS": "4"})
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    srv = None
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
        evs = ranklens.generate_requests(24, 100, N_ITEMS, N_SESS, seed=93) + ranklens.generate_requests(6, 9, N_ITEMS, N_SESS, seed=94)
        q = ranklens.column_quantiles(np.concatenate([orc.matrix(ev) for ev in evs[:6]]))
        blob = synth.synthetic_lgbm_model(n_trees=200, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.03, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [orc.rerank(ev) for ev in evs]
        reqs = [M.Request(ev) for ev in evs]
        for r in reqs[:3]:
            hip.ranker.rerank("xgboost", r, hip.booster)
        srv = hip.ranker.serve("xgboost", hip.booster, n_slots=20)
        bad, errors = [], []

        def client(t):
            try:
                for k in range(240):
                    i = (t * 5 + k) % len(reqs)
                    if (t + k) % 3 == 0:   # the host's usual entry point, answered through the started queue
                        _, s, o = hip.ranker.rerank("xgboost", reqs[i], hip.booster)
                    else:
                        s, o = srv.rerank(reqs[i])
                    if not (same(s, expected[i][1]) and o.tolist() == expected[i][2].tolist()):
                        bad.append((t, k))
                    if k % 40 == 39:
                        time.sleep(0.012 + 0.001 * (t % 4))   # beyond life and idle time, out of step: gangs are found partly left
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=client, args=(t,)) for t in range(12)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        st = srv.stats()
        print(f"\n   gangs: {st['queue']} through the queue, {st['fallback']} through the front, {st['launches']} gang launches")
        assert not errors, errors[:3]
        assert not bad, bad[:5]
        assert st["queue"] >= 12 * 240 * 0.9, st           # mrk_rank's third of the calls went through the queue too
        assert 6 <= st["launches"] <= 400, st              # revived again and again - by gangs (12 threads x 6 pauses), not per request
    finally:
        if srv is not None:
            srv.close()
        restore_env(saved)
        hip.close()


@pytest.mark.gpu
def test_busy_serving_workgroups_do_not_stall_reallocations():
    """A persistent workgroup is a resident kernel, and hipFree / a reallocation on ANY thread waits for resident kernels.
    Under sustained traffic a slot never idles (the most recently used slot is handed out first), so the workgroup must
    bound its own residency (MRK_SERVE_LIFE_US): here three threads keep two slots busy with an idle time far beyond the
    test while a fourth issues mrk_rank calls of growing size - each grows device buffers of the context.  Every such
    call must return promptly and with the oracle's bytes, and the workgroups must have been relaunched on the way."""
    import threading
    import time

    saved = with_env({"MRK_RANK_JIT": "1", "MRK_SERVE_IDLE_US": "6000000", "MRK_SERVE_LIFE_US": "3000"})
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    srv = None
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
        small = ranklens.generate_requests(8, 100, N_ITEMS, N_SESS, seed=91)
        q = ranklens.column_quantiles(np.concatenate([orc.matrix(ev) for ev in small[:4]]))
        blob = synth.synthetic_lgbm_model(n_trees=300, n_features=24, quantiles=q, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        # the kernels of the batch shapes ranked below, compiled (and left in the disk cache) by ANOTHER context: what is timed
        # here is a reallocation next to resident workgroups, not hiprtc (the kernels are keyed by this forest's view signature)
        warm = HipBackend(cfg, "xgboost")
        try:
            ranklens.load_state(warm, ranklens.generate_state(N_ITEMS, N_SESS))
            warm.load_model(blob, 0)
            for n in (129, 300, 700, 1500, 2900):
                warm.rerank(ranklens.generate_requests(1, n, N_ITEMS, N_SESS, seed=100 + n)[0])
            # ... and mrk_rank's one-launch kernel (no explain matrix): a hammer thread that finds both slots busy falls back
            # to it, and under MRK_RANK_JIT=1 its first use would compile for ~4 s with the batching front's leadership held -
            # exactly the wait this test must not mistake for a stalled reallocation
            warm.ranker.rerank("xgboost", small[0], warm.booster, explain=False)
        finally:
            warm.close()
        srv = hip.ranker.serve("xgboost", hip.booster, n_slots=2)
        exp_small = [orc.rerank(ev) for ev in small]
        srv.rerank(small[0])
        stop = threading.Event()
        bad = []

        t_end = time.perf_counter() + 8.0   # (a broken bound must fail this test, not hang the box)

        def hammer(t):
            k = t
            while not stop.is_set() and time.perf_counter() < t_end:
                s, o = srv.rerank(small[k % len(small)])
                e = exp_small[k % len(small)]
                if not (same(s, e[1]) and o.tolist() == e[2].tolist()):
                    bad.append(k)
                k += 3

        threads = [threading.Thread(target=hammer, args=(t,)) for t in range(3)]
        for t in threads:
            t.start()
        worst = 0.0
        try:
            time.sleep(0.05)
            for n in (129, 300, 700, 1500, 2900):   # each larger than the last: d_in / d_cells / d_sort grow every time
                ev = ranklens.generate_requests(1, n, N_ITEMS, N_SESS, seed=100 + n)[0]
                _, es, eo = orc.rerank(ev)
                t0 = time.perf_counter()
                _, s, o = hip.rerank(ev)
                worst = max(worst, time.perf_counter() - t0)
                assert same(s, es) and o.tolist() == eo.tolist(), n
        finally:
            stop.set()
            for t in threads:
                t.join()
        assert not bad
        st = srv.stats()
        assert worst < 2.0, (worst, st)          # without the bound: as long as the traffic lasts (here: for ever)
        assert st["launches"] > 4, st            # the workgroups did leave and come back
    finally:
        if srv is not None:
            srv.close()
        restore_env(saved)
        hip.close()


@pytest.mark.gpu
def test_default_jit_mode_never_waits_for_the_compiler():
    """MRK_RANK_JIT=auto (the library's default): (a) the stock Ranklens program's kernels are shipped next to the library
    (metarank_amd/jit_cache, built by __graft_entry__.build()): the first mrk_rank of a fresh context takes milliseconds and
    already runs them; (b) a program nobody has compiled is ranked by the interpreting kernel at once while its kernel
    compiles in the background - same bytes before and after the swap."""
    import tempfile
    import time

    saved = with_env({"MRK_RANK_JIT": "auto", "MRK_JIT_CACHE_DIR": tempfile.mkdtemp(prefix="mrk_jit_test_")})   # an empty user cache
    try:
        shipped = os.path.join(os.path.dirname(M.__file__), "jit_cache")
        assert os.path.isdir(shipped) and len([f for f in os.listdir(shipped) if f.endswith(".co")]) >= 6, "run __graft_entry__.build()"
        for variant in ("stock", "new"):
            cfg = ranklens.ranklens_config()
            if variant == "new":   # one feature less: a program no cache has seen
                cfg["features"] = [f for f in cfg["features"] if f["name"] != "runtime"]
                cfg["models"]["xgboost"]["features"] = [f for f in cfg["models"]["xgboost"]["features"] if f != "runtime"]
            orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
            try:
                for b in (orc, hip):
                    ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
                reqs = ranklens.generate_requests(4, 100, N_ITEMS, N_SESS, seed=86)
                dim = hip.dim
                blob = synth.synthetic_lgbm_model(n_trees=300, n_features=dim, quantiles=ranklens.column_quantiles(np.concatenate([orc.matrix(ev) for ev in reqs])))
                orc.load_model(blob, 0)
                hip.load_model(blob, 0)
                hip.ranker.flush()
                t0 = time.perf_counter()
                _, s, o = hip.ranker.rerank("xgboost", reqs[0], hip.booster)
                first_ms = (time.perf_counter() - t0) * 1e3
                _, es, eo = orc.rerank(reqs[0])
                assert same(s, es) and o.tolist() == eo.tolist()
                # a hiprtc compile of this kernel takes 3 - 20 s depending on the host; what the first request does pay since
                # round 4 is the loading of the program-only kernels that are on disk (seven modules, before the background
                # compile starts: jit.cpp) - milliseconds on a fast host, more on the pool's slow ones
                assert first_ms < 1000.0, (variant, first_ms)
                hip.ranker.warmup_kernels("xgboost")            # the background compile (variant "new") is done after this
                for ev in reqs:
                    _, s, o = hip.ranker.rerank("xgboost", ev, hip.booster)
                    _, es, eo = orc.rerank(ev)
                    assert same(s, es) and o.tolist() == eo.tolist(), variant
            finally:
                hip.close()
    finally:
        restore_env(saved)


@pytest.mark.gpu
def test_kernel_keys_say_which_specialised_kernels_ran():
    """mrk_config_kernel_keys: after a batch has run with the compiler waited for, the model's fused kernel is listed with the key
    of its translation unit - keyed by program AND forest by default, by the program only under MRK_JIT_SIG=0 - and the two
    keys differ (bench.py's `provenance`)."""
    cfg = ranklens.ranklens_config()
    keys = {}
    for sig in ("1", "0"):
        saved = with_env({"MRK_RANK_JIT": "1", "MRK_JIT_SIG": sig})
        hip = HipBackend(cfg, "xgboost")
        try:
            ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
            hip.load_model(synth.synthetic_lgbm_model(n_trees=50, n_features=24, missing="per_feature"), 0)
            batch = hip.ranker.prepare("xgboost", ranklens.generate_requests(80, 100, N_ITEMS, N_SESS, seed=7))   # (more than 64 small requests: the unsplit kernel)
            batch.run(hip.booster)
            batch.fetch()
            batch.close()
            got = hip.ranker.kernel_keys("xgboost")
            assert list(got) == ["mrk_jit_rank_cells"] and len(got["mrk_jit_rank_cells"]) == 1, got
            keys[sig] = got["mrk_jit_rank_cells"][0]
        finally:
            hip.close()
            restore_env(saved)
    assert keys["1"].endswith(" program+forest") and keys["0"].endswith(" program") and keys["1"].split()[0] != keys["0"].split()[0]

