"""Randomised parity of the whole hot path on synthetic Ranklens-shaped state: the HIP path
(through the C ABI) against the CPU oracle — dense matrix bit-exact, scores bit-exact (well inside
the 1e-5 of BASELINE.json), response order identical."""
import numpy as np
import pytest

from backends import HipBackend, OracleBackend
import metarank_amd as M
from workloads import ranklens, synth

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
@pytest.mark.parametrize("kind", ["lgbm", "xgb"])
def test_forests_with_long_threshold_tables(oracle_c2, kind):
    """Round 6: the assembly kernels keep a forest's threshold tables resident in LDS when they are small (the benchmark's model:
    ~50 distinct thresholds per column).  Forests that are not: ~250 thresholds per column (the tables no longer fit the
    workgroup-per-request kernels' budget: the staging sinks run; the item-parallel kernel still holds them - 40 KB) and columns
    with MORE than 256 thresholds (never resident, never staged: searched in global memory) next to short ones.  Scores and
    order equal the oracle's through the one-launch kernel, the batch kernels and the item-parallel kernel (300-item requests)."""
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        load(hip)
        reqs = ranklens.generate_requests(24, 100, N_ITEMS, N_SESS, seed=31) + ranklens.generate_requests(3, 300, N_ITEMS, N_SESS, seed=32)
        mats = [oracle_c2.matrix(ev) for ev in reqs]
        q = ranklens.column_quantiles(np.concatenate(mats), n=900)
        assert max(len(c) for c in q) > 300
        if kind == "lgbm":   # 7 500 splits over 24 columns: up to ~250 distinct thresholds on the continuous columns
            blob, be = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.02, missing="per_feature"), 0
        else:                # 12 000 splits (16-leaf trees: the bit-vector scorer): the continuous columns beyond 256
            blob, be = synth.synthetic_xgb_model(n_trees=800, n_features=24, depth=4, quantiles=q), 1
        oracle_c2.load_model(blob, be)
        hip.load_model(blob, be)
        assert hip.booster.info()["bitvector"] == 1
        expected = [oracle_c2.rerank(ev) for ev in reqs]
        for ev, (_, es, eo) in list(zip(reqs, expected))[:6] + [(reqs[-1], expected[-1])]:
            _, hs, ho = hip.rerank(ev)
            assert same(hs, es) and ho.tolist() == eo.tolist()
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, _ = batch.fetch()
        assert (batch.status() == 0).all()
        for r, (_, es, eo) in enumerate(expected):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), r
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


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lgbm", "xgb4"])
def test_assembly_paths_agree(oracle_c2, kind):
    """The four ways a batch can be assembled - {one fused workgroup per request with LDS tables,
    pre-pass kernel + item-parallel kernel with HBM tables} x {straight into the scorer's binned tile,
    f64 matrix + binning kernel} - give the oracle's scores and order; item-field overrides and a request
    that throws are in the batch."""
    import os

    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    saved = {k: os.environ.get(k) for k in ("MRK_RANK_FUSED", "MRK_RANK_CELLS", "MRK_RANK_JIT", "MRK_JIT_SIG", "MRK_FUSED_SPLIT", "MRK_FUSED_SLICES", "MRK_FUSED_THREADS", "MRK_ITEMS_LDS")}
    try:
        load(hip)
        reqs = ranklens.generate_requests(30, 100, N_ITEMS, N_SESS, seed=21)
        reqs += ranklens.generate_requests(2, 1, N_ITEMS, N_SESS, seed=22) + ranklens.generate_requests(2, 300, N_ITEMS, N_SESS, seed=23)
        reqs.append({"id": "empty-ish", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": [{"id": "nobody"}]})
        # per-request inputs that win over the store (NumberFeature.scala:86-92, StringFeature.scala:96-99)
        for ev in reqs[:6]:
            ev["items"][0]["fields"] = [{"name": "popularity", "value": 77.5}, {"name": "genres", "value": ["drama", "comedy"]}]
            ev["items"][3]["fields"] = [{"name": "vote_avg", "value": float("nan")}]
        mats = [oracle_c2.matrix(ev) for ev in reqs]
        q = ranklens.column_quantiles(np.concatenate(mats))
        if kind == "lgbm":
            blob, be = synth.synthetic_lgbm_model(n_trees=300, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.03), 0
        else:
            blob, be = synth.synthetic_xgb_model(n_trees=200, n_features=24, depth=4, quantiles=q, cat_features=[7], cat_prob=0.05), 1
        oracle_c2.load_model(blob, be)
        hip.load_model(blob, be)
        assert hip.booster.info()["bitvector"] == 1
        expected = [oracle_c2.rerank(ev) for ev in reqs]
        # (fused, cells, jit): the hot path runs the kernel specialised for this model's feature list at run time
        # ("require": a hiprtc failure is an error, not a fall-back); "0" = the generic kernel that interprets the program
        # cells "0" + jit "require": the specialised kernel's f64-matrix form (the path of models scored by the tree walk)
        # sig "0": the specialised kernel keyed by the program only (it reads the forest's column descriptors from memory - what
        # serves a retrained model while the kernel keyed by its view signature compiles); default: keyed by both
        for fused, cells, jit, sig in (("1", "1", "require", "1"), ("1", "1", "require", "0"), ("1", "1", "0", "1"), ("1", "0", "require", "1"), ("1", "0", "0", "1"),
                                       ("0", "1", "require", "1"), ("0", "1", "require", "0"), ("0", "1", "0", "1"), ("0", "0", "0", "1")):
            if True:
                os.environ["MRK_RANK_FUSED"], os.environ["MRK_RANK_CELLS"], os.environ["MRK_RANK_JIT"], os.environ["MRK_JIT_SIG"] = fused, cells, jit, sig
                M.reload_switches()
                batch = hip.ranker.prepare("xgboost", reqs)
                batch.run(hip.booster)
                scores, order, _ = batch.fetch()
                assert (batch.status() == 0).all()
                for r, (_, es, eo) in enumerate(expected):
                    lo, hi = batch.offsets[r], batch.offsets[r + 1]
                    assert same(scores[lo:hi], es), (fused, cells, jit, r)
                    assert order[lo:hi].tolist() == eo.tolist(), (fused, cells, jit, r)
                # the matrix is materialised on demand and is the oracle's
                _, _, mat = batch.fetch(matrix=True)
                for r in range(len(reqs)):
                    assert same(mat[batch.offsets[r]:batch.offsets[r + 1]], mats[r]), (fused, cells, r)
                # and the next run (straight into the tile again) still gives the same scores
                batch.run(hip.booster)
                s2, o2, _ = batch.fetch()
                assert same(s2, scores) and (o2 == order).all()
                batch.close()
        # the item-parallel kernel probing the tables in the HBM arena only (default: a workgroup whose lanes share one request
        # copies that request's tables into its LDS first - the 300-candidate requests have such workgroups, the others straddle)
        for jit in ("require", "0"):
            os.environ.update(MRK_RANK_FUSED="0", MRK_RANK_CELLS="1", MRK_RANK_JIT=jit, MRK_ITEMS_LDS="0")
            M.reload_switches()
            batch = hip.ranker.prepare("xgboost", reqs)
            batch.run(hip.booster)
            scores, order, _ = batch.fetch()
            assert (batch.status() == 0).all()
            for r, (_, es, eo) in enumerate(expected):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), ("items, arena tables", jit, r)
            batch.close()
        os.environ.pop("MRK_ITEMS_LDS", None)
        # round 6: the two sinks of the signature-keyed kernels - compact tables RESIDENT in LDS, several searches at a time (the default
        # where they fit) vs a table STAGED per wavefront and column - in the workgroup-per-request kernel, its split form and the
        # item-parallel kernel (persistent resident-table form vs the staging form): identical bytes
        saved_defs = os.environ.get("MRK_JIT_DEFINES")
        for fused, split, defs, items_rt in (("1", None, "MRK_FUSED_RT_MAX=0 MRK_FUSED_RT_MAX_SPLIT=0", "1"), ("1", "2", "MRK_FUSED_RT_MAX=0 MRK_FUSED_RT_MAX_SPLIT=0", "1"),
                                             ("1", "2", None, "1"), ("0", None, None, "0"), ("0", None, "MRK_RT_Q=2", "1")):
            os.environ.update(MRK_RANK_FUSED=fused, MRK_RANK_CELLS="1", MRK_RANK_JIT="require", MRK_JIT_SIG="1", MRK_ITEMS_RT=items_rt)
            for k, v in (("MRK_FUSED_SPLIT", split), ("MRK_JIT_DEFINES", defs)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            M.reload_switches()
            batch = hip.ranker.prepare("xgboost", reqs)
            batch.run(hip.booster)
            scores, order, _ = batch.fetch()
            assert (batch.status() == 0).all()
            for r, (_, es, eo) in enumerate(expected):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), ("sinks", fused, split, defs, items_rt, r)
            batch.close()
        for k, v in (("MRK_JIT_DEFINES", saved_defs), ("MRK_ITEMS_RT", None), ("MRK_FUSED_SPLIT", None)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        # op split (what a handful of requests gets by itself): the item lanes in 2 / 4 copies that share the ops,
        # specialised kernel (tile), interpreting kernels (tile and f64 matrix)
        for split in ("2", "4"):
            for cells, jit in (("1", "require"), ("1", "0"), ("0", "0")):
                os.environ["MRK_FUSED_SPLIT"], os.environ["MRK_RANK_FUSED"], os.environ["MRK_RANK_CELLS"], os.environ["MRK_RANK_JIT"] = split, "1", cells, jit
                M.reload_switches()
                batch = hip.ranker.prepare("xgboost", reqs)
                batch.run(hip.booster)
                scores, order, mat = batch.fetch(matrix=True)
                assert (batch.status() == 0).all()
                for r, (_, es, eo) in enumerate(expected):
                    lo, hi = batch.offsets[r], batch.offsets[r + 1]
                    assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), (split, cells, jit, r)
                    assert same(mat[lo:hi], mats[r]), (split, cells, jit, r)
                batch.close()
        os.environ.pop("MRK_FUSED_SPLIT", None)
        # slices (what a batch of few LARGE requests gets by itself): several workgroups per request, each with its own
        # pre-pass tables and a slice of the candidates; with 64 item lanes the 300-candidate requests take 5 rounds, so
        # up to 5 slices, the 100-candidate ones 2, the single-candidate ones 1 (the other workgroups leave at once)
        for slices, split in (("2", "1"), ("5", "1"), ("3", "2")):
            for cells, jit in (("1", "require"), ("1", "0"), ("0", "0")):
                os.environ.update(MRK_FUSED_SLICES=slices, MRK_FUSED_SPLIT=split, MRK_FUSED_THREADS="64", MRK_RANK_FUSED="1", MRK_RANK_CELLS=cells, MRK_RANK_JIT=jit)
                M.reload_switches()
                batch = hip.ranker.prepare("xgboost", reqs)
                batch.run(hip.booster)
                scores, order, mat = batch.fetch(matrix=True)
                assert (batch.status() == 0).all()
                for r, (_, es, eo) in enumerate(expected):
                    lo, hi = batch.offsets[r], batch.offsets[r + 1]
                    assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), (slices, split, cells, jit, r)
                    assert same(mat[lo:hi], mats[r]), (slices, split, cells, jit, r)
                batch.close()
        for k in ("MRK_FUSED_SLICES", "MRK_FUSED_SPLIT", "MRK_FUSED_THREADS"):
            os.environ.pop(k, None)
        # single requests: mrk_rank without / with the explain matrix
        os.environ["MRK_RANK_FUSED"], os.environ["MRK_RANK_CELLS"], os.environ["MRK_RANK_JIT"] = "1", "1", "require"
        M.reload_switches()
        for ev, (_, es, eo), m in list(zip(reqs, expected, mats))[:8]:
            _, hs, ho = hip.ranker.rerank("xgboost", ev, hip.booster, explain=False)
            assert same(hs, es) and ho.tolist() == eo.tolist()
            hm, hs, ho = hip.ranker.rerank("xgboost", ev, hip.booster, explain=True)
            assert same(hm, m) and same(hs, es) and ho.tolist() == eo.tolist()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        M.reload_switches()
        hip.close()


@pytest.mark.gpu
def test_c4_100k_candidates_sharded_and_sorted(oracle_c2):
    """BASELINE config C4: one request with 100 000 candidates.  (a) whole on one GPU: multi-workgroup
    assembly, HBM pre-pass tables, multi-workgroup sort; (b) item-sharded 2 and 8 ways the way 8 GPUs
    would run it (each shard assembles + scores its slice, the slices are merged, then one sort):
    same bytes as the oracle's rerank of the whole request."""
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        load(hip)
        big = ranklens.generate_requests(1, 100_000, N_ITEMS, N_SESS, seed=41)[0]
        small = ranklens.generate_requests(3, 100, N_ITEMS, N_SESS, seed=42)
        reqs = [small[0], big, small[1], small[2]]
        sample = np.concatenate([oracle_c2.matrix(ev) for ev in small])
        blob = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=ranklens.column_quantiles(sample),
                                          cat_features=[7], cat_prob=0.01, missing="per_feature")
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [oracle_c2.rerank(ev) for ev in reqs]
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, _ = batch.fetch()
        assert (batch.status() == 0).all()
        for r, (_, es, eo) in enumerate(expected):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            assert same(scores[lo:hi], es), r
            assert order[lo:hi].tolist() == eo.tolist(), r
        # ties and NaN in a large sort: stable order == argsort(kind="stable") on the Double.compare key
        assert len(np.unique(scores[batch.offsets[1]:batch.offsets[2]])) < 100_000  # duplicates exist: the tie-break matters
        for world in (2, 8):
            chunk = batch.shard_chunk(world)
            assert chunk % 128 == 0 and chunk * world >= batch.total_items
            b2 = hip.ranker.prepare("xgboost", reqs)
            for rank in range(world):  # same device buffer: the slices land where an all-gather would put them
                b2.run_shard(hip.booster, rank, world)
            b2.sort()
            s2, o2, _ = b2.fetch()
            assert same(s2, scores) and (o2 == order).all(), world
            b2.close()
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["noop", "linear", "position"])
def test_c5_biencoder_column(norm):
    """BASELINE config C5 minus the ONNX forward (the query embedding arrives as a request field): 24 Ranklens
    columns + the bi-encoder cosine column (f32 query x f64 item, f64 accumulators, no epsilon,
    FieldMatchBiencoderFeature.scala:80-109 / DistanceFunction.scala:14-26), then schema.norm.scale over the request's
    column (ml/onnx/Normalize.scala:13-45: noop | linear = min-max | position = rank / size), 500-tree LambdaMART."""
    cfg = ranklens.c5_config()
    cfg["features"][-1]["norm"] = norm
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for b in (orc, hip):
            load(b)
            ranklens.load_state(b, ranklens.c5_embeddings(N_ITEMS))
        assert orc.dim == 25 and hip.dim == 25
        reqs = ranklens.generate_requests(12, 100, N_ITEMS, N_SESS, seed=51)
        for k, ev in enumerate(reqs):
            if k % 4 != 3:  # every fourth request has no query: the column is NaN for all its items
                ev["fields"] = [{"name": "__embedding:title_match", "value": ranklens.c5_query(seed=k)}]
        reqs[0]["items"][0]["id"] = "7"  # the all-zero embedding: 0 / (x * 0) = NaN
        reqs[1]["items"][5] = dict(reqs[1]["items"][4])  # the same item twice: equal cosines (a tie for the position rank)
        reqs.append(dict(reqs[2], id="one", items=reqs[2]["items"][:1]))  # a single candidate: linear gives 0 / 0 = NaN
        mats = [orc.matrix(ev) for ev in reqs]
        col = np.concatenate(mats)[:, 24]
        assert np.isfinite(col).any() and np.isnan(col).any() and np.nanmax(np.abs(col)) <= 1.0 + 1e-12
        if norm != "noop":
            assert np.nanmin(col) == 0.0 and (norm == "position" or np.nanmax(col) == 1.0)
        blob = synth.synthetic_lgbm_model(n_trees=500, n_features=25, quantiles=ranklens.column_quantiles(np.concatenate(mats)),
                                          cat_features=[7], cat_prob=0.01, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        scores, order, mat = batch.fetch(matrix=True)
        assert (batch.status() == 0).all()
        for r, ev in enumerate(reqs):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            _, es, eo = orc.rerank(ev)
            assert same(mat[lo:hi], mats[r]), r
            assert same(scores[lo:hi], es), r
            assert order[lo:hi].tolist() == eo.tolist(), r
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_concurrent_callers_are_combined_and_isolated(oracle_c2):
    """The batching front of mrk_rank: 16 threads rank different requests at the same time (the reference's
    threading model: one rerank per request thread); every caller gets exactly its own request's scores and
    order, the explain matrix when it asked for it, and a request that throws fails alone."""
    from concurrent.futures import ThreadPoolExecutor

    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        load(hip)
        reqs = ranklens.generate_requests(96, 100, N_ITEMS, N_SESS, seed=61) + ranklens.generate_requests(8, 7, N_ITEMS, N_SESS, seed=62)
        mats = [oracle_c2.matrix(ev) for ev in reqs]
        blob = synth.synthetic_lgbm_model(n_trees=200, n_features=24, quantiles=ranklens.column_quantiles(np.concatenate(mats)),
                                          cat_features=[7], cat_prob=0.02)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [oracle_c2.rerank(ev) for ev in reqs]

        def one(k):
            explain = k % 5 == 0
            m, s, o = hip.ranker.rerank("xgboost", reqs[k], hip.booster, explain=explain)
            return k, m, s, o

        for _ in range(3):
            with ThreadPoolExecutor(16) as ex:
                results = list(ex.map(one, range(len(reqs))))
            for k, m, s, o in results:
                em, es, eo = expected[k]
                assert same(s, es) and o.tolist() == eo.tolist(), k
                assert m is None or same(m, em), k
        # a request that makes the reference throw (normalised rate, global clicks == 0) among healthy ones
        for b in (hip,):
            b.put_periodic("global/ctr_click_norm", [0, 5])

        def guarded(k):
            try:
                hip.ranker.rerank("xgboost", reqs[k], hip.booster)
                return None
            except Exception as e:  # noqa: BLE001
                return e

        with ThreadPoolExecutor(8) as ex:
            errs = list(ex.map(guarded, range(24)))
        assert all(e is not None and getattr(e, "status", 0) == -5 for e in errs)  # every request reads the global counter
    finally:
        hip.close()


def _stress_config():
    """Shapes the Ranklens workloads do not reach: rates over 6 periods (more than one batch of the rate op), an
    interacted_with over 6 fields (more than one batch of fields), token lists of up to 14 tokens (beyond the batches the
    kernels prefetch), diversity over 40 numeric values."""
    periods = [1, 2, 3, 7, 14, 30]
    fields = [f"f{k}" for k in range(6)]
    features = [
        {"name": "pop", "type": "number", "scope": "item", "source": "metadata.pop"},
        {"name": "price", "type": "number", "scope": "item", "source": "metadata.price"},
        {"name": "ctr6", "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h", "periods": periods, "normalize": {"weight": 3}},
        {"name": "ctr_tag6", "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h", "periods": periods, "scope": "item.tag"},
        {"name": "profile6", "type": "interacted_with", "interaction": "click", "field": [f"item.{f}" for f in fields], "field_order": fields,
         "scope": "session", "count": 100, "duration": "24h"},
        {"name": "div_words", "type": "diversity", "source": "item.words", "top": 30},
        {"name": "div_price", "type": "diversity", "source": "item.price", "top": 40},
        {"name": "clicks", "type": "window_count", "interaction": "click", "scope": "item", "bucket": "24h", "periods": periods},
    ]
    names = [f["name"] for f in features]
    return {"features": features, "models": {"m": {"type": "lambdamart", "backend": {"type": "lightgbm", "iterations": 10}, "features": names}}}, periods, fields


def _stress_state(periods, fields, n_items=600, n_sessions=40, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    P = len(periods)
    tag_tot = {}
    g = [np.zeros(P, dtype=np.int64), np.zeros(P, dtype=np.int64)]
    for i in range(n_items):
        k = f"item={i}/"
        if rng.random() < 0.9:
            yield "double", k + "pop", float(rng.integers(0, 100000))
            yield "double", k + "price", float(np.round(rng.normal() * 50, 2))
            yield "double", k + "div_price", float(np.round(rng.normal() * 50, 2)) if rng.random() < 0.95 else float("nan")
            for f in fields:
                yield "string_list", k + f"profile6_{f}", [f"{f}t{t}" for t in rng.integers(0, 60, int(rng.integers(0, 15)))]
            yield "string_list", k + "div_words", [f"w{t}" for t in rng.integers(0, 200, int(rng.integers(1, 13)))]
        if rng.random() < 0.85:
            imp = np.sort(rng.integers(1, 5000, P))
            clk = (imp * rng.random(P) * 0.3).astype(np.int64)
            yield "periodic", k + "ctr6_click", [int(x) for x in clk]
            yield "periodic", k + "ctr6_impression", [int(x) for x in imp]
            yield "periodic", k + "clicks", [int(x) for x in clk]
            g[0] += clk
            g[1] += imp
            if rng.random() < 0.7:
                tag = f"tag{int(rng.integers(0, 25))}"
                yield "string", k + "ctr_tag6_field", tag
                e = tag_tot.setdefault(tag, [np.zeros(P, dtype=np.int64), np.zeros(P, dtype=np.int64)])
                e[0] += clk
                e[1] += imp
    yield "periodic", "global/ctr6_click_norm", [int(x) for x in g[0]]
    yield "periodic", "global/ctr6_impression_norm", [int(x) for x in g[1]]
    for tag, (c, m) in tag_tot.items():
        yield "periodic", f"field=tag:{tag}/ctr_tag6_click", [int(x) for x in c]
        yield "periodic", f"field=tag:{tag}/ctr_tag6_impression", [int(x) for x in m]
    for s in range(n_sessions):
        ln = int(rng.integers(0, 101))
        if ln:
            yield "bounded_list", f"session=s{s}/profile6_interactions", [str(x) for x in rng.integers(0, n_items, ln)]


@pytest.mark.gpu
def test_shapes_beyond_the_kernels_batches_and_long_threshold_tables():
    """Rates over 6 periods, interacted_with over 6 fields, token lists beyond the prefetched batches, a 40-value numeric
    diversity - and a forest whose columns carry up to > 256 distinct thresholds, so the assembly sink stages tables in one
    LDS-DMA chunk, in two, and falls back to searching in global memory.  Specialised and generic kernels vs the oracle."""
    import os

    cfg, periods, fields = _stress_config()
    orc, hip = OracleBackend(cfg, "m"), HipBackend(cfg, "m")
    saved = os.environ.get("MRK_RANK_JIT")
    try:
        for be in (orc, hip):
            ranklens.load_state(be, _stress_state(periods, fields))
        reqs = ranklens.generate_requests(24, 100, 600, 40, seed=31) + ranklens.generate_requests(3, 300, 600, 40, seed=32)
        mats = [orc.matrix(ev) for ev in reqs]
        allm = np.concatenate(mats)
        dim = allm.shape[1]
        assert dim == 2 + 6 + 6 + 6 + 1 + 1 + 6
        # split candidates per column: 40 / 200 / 400 quantiles -> tables of < 128, 129..256 and > 256 thresholds
        q = []
        for j in range(dim):
            col = allm[:, j]
            col = col[np.isfinite(col)]
            n = (40, 200, 400)[j % 3]
            q.append(np.unique(np.quantile(col, np.linspace(0.01, 0.99, n))) if len(col) else np.array([0.0, 0.5]))
        blob = synth.synthetic_lgbm_model(n_trees=3000, n_features=dim, quantiles=q, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        assert hip.booster.info()["bitvector"] == 1
        expected = [orc.rerank(ev) for ev in reqs]
        for jit in ("require", "0"):
            os.environ["MRK_RANK_JIT"] = jit
            M.reload_switches()
            batch = hip.ranker.prepare("m", reqs)
            batch.run(hip.booster)
            scores, order, _ = batch.fetch()
            assert (batch.status() == 0).all()
            for r, (_, es, eo) in enumerate(expected):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es), (jit, r)
                assert order[lo:hi].tolist() == eo.tolist(), (jit, r)
            _, _, mat = batch.fetch(matrix=True)
            for r in range(len(reqs)):
                assert same(mat[batch.offsets[r]:batch.offsets[r + 1]], mats[r]), (jit, r)
            batch.close()
    finally:
        if saved is None:
            os.environ.pop("MRK_RANK_JIT", None)
        else:
            os.environ["MRK_RANK_JIT"] = saved
        M.reload_switches()
        hip.close()


@pytest.mark.gpu
def test_background_specialisation_swaps_in_without_changing_results(oracle_c2, tmp_path):
    """MRK_RANK_JIT=async: the first rank of a model starts the hiprtc compile on a background thread and is served by the
    generic kernel; once the code object exists (it appears in the cache directory) the specialised kernel takes over.
    Every answer on the way is the oracle's."""
    import glob
    import os
    import time

    saved = {k: os.environ.get(k) for k in ("MRK_RANK_JIT", "MRK_JIT_CACHE_DIR", "MRK_JIT_SHIPPED")}
    os.environ["MRK_RANK_JIT"], os.environ["MRK_JIT_CACHE_DIR"] = "async", str(tmp_path)
    os.environ["MRK_JIT_SHIPPED"] = "0"   # (the stock program's kernels ship next to the library: this test wants the compile)
    M.reload_switches()
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        load(hip)
        reqs = ranklens.generate_requests(12, 100, N_ITEMS, N_SESS, seed=41)
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in reqs]))
        blob = synth.synthetic_lgbm_model(n_trees=100, n_features=24, quantiles=q)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [oracle_c2.rerank(ev) for ev in reqs]

        def check():
            batch = hip.ranker.prepare("xgboost", reqs)
            batch.run(hip.booster)
            scores, order, _ = batch.fetch()
            for r, (_, es, eo) in enumerate(expected):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), r
            batch.close()

        t0 = time.time()
        check()  # generic kernel; the compile starts
        assert time.time() - t0 < 5.0, "the first rank waited for the compiler"
        deadline = time.time() + 120
        while not glob.glob(str(tmp_path / "*.co")) and time.time() < deadline:
            check()
            time.sleep(0.25)
        assert glob.glob(str(tmp_path / "*.co")), "no code object was produced"
        for _ in range(3):  # the specialised kernel is loaded by the next run and used from then on
            check()
    finally:
        hip.close()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        M.reload_switches()
