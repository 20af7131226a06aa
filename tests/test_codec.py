"""Bulk load from Metarank's binary FeatureValue wire format.  CPU: the test-side writer/reader
restatement round-trips the reference's own test values and VarNum known answers; GPU: a store loaded from the
binary blobs (mrk_store_put_binary) ranks exactly like the same store loaded through the typed puts."""
import numpy as np
import pytest

from oracle import codec


def test_varnum_known_answers():  # util/VarNum.java = unsigned LEB128 (hand-computed)
    assert codec.var_long(0) == b"\x00" and codec.var_long(1) == b"\x01" and codec.var_long(127) == b"\x7f"
    assert codec.var_long(128) == b"\x80\x01" and codec.var_long(300) == b"\xac\x02"
    assert codec.var_long(1661345221008) == bytes([0x90, 0x93, 0x91, 0xff, 0xac, 0x30])
    assert codec.var_long(-1) == b"\xff" * 9 + b"\x01"  # two's complement: ten bytes
    assert codec.var_int(300) == b"\xac\x02" and codec.var_int(-1) == b"\xff\xff\xff\xff\x0f"
    assert codec.utf("foo") == b"\x00\x03foo" and codec.utf("é€") == b"\x00\x05\xc3\xa9\xe2\x82\xac"
    assert codec.utf("a\x00b") == b"\x00\x04a\xc0\x80b"  # modified UTF-8
    assert codec.f64(1.0) == b"\x3f\xf0" + b"\x00" * 6


def test_varnum_is_protobuf_varint():
    """util/VarNum.java's layout is the protobuf base-128 varint of the 64-bit two's complement; protobuf's own encoder
    (an implementation that shares nothing with oracle/codec.py or csrc/codec.cpp) writes the same bytes."""
    import random

    from google.protobuf.internal.encoder import _VarintBytes

    rnd = random.Random(1)
    for _ in range(5000):
        v = rnd.getrandbits(rnd.randint(1, 63)) * rnd.choice((1, 1, -1))
        assert codec.var_long(v) == _VarintBytes(v & (2**64 - 1)), v
        if -2**31 <= v < 2**31:
            assert codec.var_int(v) == _VarintBytes(v & (2**32 - 1)), v


def test_roundtrip_of_the_reference_test_values():  # T/fstore/redis/codec/impl/FeatureValueCodecTest.scala:27-53
    k, ts = "user=u1/foo", 1661345221008
    values = [("string", "foo"), ("counter", 1), ("numstats", (1.0, 2.0, {1: 1.0})), ("map", {"foo": ("string", "bar")}),
              ("periodic", [1]), ("freq", {"foo": 1.0}), ("bounded_list", ["foo"])]
    for compat in (False, True):
        blob = b"".join(codec.feature_value(kind, k, v, ts, compat=compat) for kind, v in values)
        back = codec.decode(blob)
        assert [(b[0], b[1], b[2], b[3]) for b in back] == [(kind, k, v, ts) for kind, v in values]
        assert all(b[4] == (None if compat else codec.DAYS_90_MS) for b in back)
    # every scope of ScopeCodec
    for key in ("item=i/x", "global/x", "session=s/x", "field=genre:drama/x", "irf=query:red socks:p1/x", "ranking=r1/x"):
        assert codec.decode(codec.feature_value("double", key, 2.5))[0][1:3] == (key, 2.5)
    assert codec.decode(codec.feature_value("double_list", "item=i/v", [1.5, -2.0]))[0][2] == [1.5, -2.0]
    assert codec.decode(codec.feature_value("string_list", "item=i/v", ["a", "é"]))[0][2] == ["a", "é"]
    assert codec.decode(codec.feature_value("counter", "item=i/c", -5))[0][2] == -5


@pytest.mark.gpu
def test_store_loaded_from_binary_ranks_like_typed_puts():
    from backends import HipBackend
    from workloads import ranklens, synth

    cfg = ranklens.c3_config()
    a, b = HipBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        state = list(ranklens.generate_state(2000, 200, c3=True))
        ranklens.load_state(a, iter(state))
        blob = bytearray()
        for i, (kind, key, value) in enumerate(state):
            blob += codec.feature_value(kind, key, value, compat=(i % 7 == 0))
            if i % 500 == 0:  # values /rank does not read must be skipped cleanly
                blob += codec.feature_value("numstats", key, (0.0, 1.0, {50: 0.5}))
                blob += codec.feature_value("map", key, {"k": ("double", 1.0)}) + codec.feature_value("freq", key, {"k": 0.5})
        n = b.ranker.put_binary(bytes(blob))
        assert n == len(state) + 3 * len(range(0, len(state), 500))
        reqs = ranklens.generate_requests(8, 300, 2000, 200, seed=7)
        ma = np.concatenate([a.matrix(ev) for ev in reqs])
        mb = np.concatenate([b.matrix(ev) for ev in reqs])
        assert ((ma == mb) | (np.isnan(ma) & np.isnan(mb))).all()
        # ... and like the ORACLE given the same values through its typed puts: the binary-loaded device store against an
        # implementation that shares nothing with it (not only HIP against HIP)
        from backends import OracleBackend

        o = OracleBackend(cfg, "xgboost")
        ranklens.load_state(o, iter(state))
        mo = np.concatenate([o.matrix(ev) for ev in reqs])
        assert ((mb == mo) | (np.isnan(mb) & np.isnan(mo))).all()
        model = synth.synthetic_lgbm_model(n_trees=100, n_features=64, quantiles=ranklens.column_quantiles(ma))
        a.load_model(model, 0)
        b.load_model(model, 0)
        for ev in reqs:
            (_, sa, oa), (_, sb, ob) = a.rerank(ev), b.rerank(ev)
            assert np.array_equal(sa, sb) and oa.tolist() == ob.tolist()
        with pytest.raises(a.M.MrkError):
            b.ranker.put_binary(bytes(blob[:-3]))  # truncated record
        with pytest.raises(a.M.MrkError):
            b.ranker.put_binary(b"\x63")           # "cannot decode fv index"
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_values_expire_after_their_ttl_like_the_redis_store():
    """FeatureValue.expire (FeatureValueCodec.scala:42-48,75; RedisKVStore.scala:40 sets the key's TTL at every write): records
    loaded with mrk_store_put_binary_at carry deadline = now + expire; mrk_store_expire(now) drops what is overdue - exactly a
    delete: the device sees a missing value after the next flush, the ORACLE given the surviving values only agrees bit for bit;
    a rewrite moves the deadline; records of the plain mrk_store_put_binary never expire."""
    from backends import HipBackend, OracleBackend
    from workloads import ranklens

    cfg = ranklens.ranklens_config()
    b, o = HipBackend(cfg, "xgboost"), OracleBackend(cfg, "xgboost")
    try:
        state = list(ranklens.generate_state(400, 40))
        T0 = 1_700_000_000_000
        short = [(k_, key, v) for (k_, key, v) in state if key.endswith("/popularity") or key.endswith("/profile_genres")]
        rest = [x for x in state if x not in short]
        blob_short = b"".join(codec.feature_value(k_, key, v, ttl_ms=60_000) for k_, key, v in short)
        blob_rest = b"".join(codec.feature_value(k_, key, v, ttl_ms=codec.DAYS_90_MS, compat=(i % 5 == 0)) for i, (k_, key, v) in enumerate(rest))
        assert b.ranker.put_binary(blob_short, now_ms=T0) == len(short)
        assert b.ranker.put_binary(blob_rest[: len(blob_rest)], now_ms=T0) == len(rest)
        reqs = ranklens.generate_requests(6, 100, 400, 40, seed=17)
        ranklens.load_state(o, iter(state))
        same = lambda x, y: bool(((x == y) | (np.isnan(x) & np.isnan(y))).all())
        assert all(same(b.matrix(ev), o.matrix(ev)) for ev in reqs)
        assert b.ranker.expire(T0 + 59_999) == 0
        # one key rewritten 30 s later: its minute starts again
        k_, key, v = short[0]
        b.ranker.put_binary(codec.feature_value(k_, key, v, ttl_ms=60_000), now_ms=T0 + 30_000)
        assert b.ranker.expire(T0 + 60_000) == len(short) - 1
        o2 = OracleBackend(cfg, "xgboost")
        ranklens.load_state(o2, iter(rest + [short[0]]))
        assert all(same(b.matrix(ev), o2.matrix(ev)) for ev in reqs)
        assert not all(same(b.matrix(ev), o.matrix(ev)) for ev in reqs)      # something the requests read did expire
        assert b.ranker.expire(T0 + 90_000) == 1
        assert b.ranker.expire(T0 + 89 * 86_400_000) == 0                      # the 90-day values (and the pre-ttl encodings) are still there
        n_rest = b.ranker.expire(T0 + 91 * 86_400_000)
        assert 0.95 * len(rest) <= n_rest <= len(rest)                          # (state of a feature the stock config does not use is never stored, hence never tracked)
        o2.close()
    finally:
        b.close()
        o.close()


def test_ranking_event_format_rejects_garbage_without_a_gpu():
    import ctypes as C

    from metarank_amd import _native as N

    L = N.lib()
    n = C.c_int(0)
    assert L.mrk_rank_binary(None, None, b"m", b"\x00\x05ab", 4, C.byref(n), None, None, 0) == N.ERR_PARSE  # truncated UTF
    ev = codec.ranking_event({"id": "r", "timestamp": 5, "user": "u", "session": None, "fields": [], "items": ["a", "b"]})
    assert ev == b"\x00\x01r" + b"\x00" * 7 + b"\x05" + b"\x01\x00\x01u" + b"\x00" + b"\x00\x00\x00\x00" + b"\x00\x00\x00\x02" + \
        b"\x00\x01a\x00\x00\x00\x00" + b"\x00\x01b\x00\x00\x00\x00"
    # decodes (n_items is reported) but there is no context to rank with
    assert L.mrk_rank_binary(None, None, b"m", ev, len(ev), C.byref(n), None, None, 2) == N.ERR_INVALID_ARG and n.value == 2


@pytest.mark.gpu
def test_binary_requests_and_container_warmup():
    from backends import HipBackend
    from workloads import ranklens, synth

    cfg = ranklens.ranklens_config()
    hip = HipBackend(cfg, "xgboost")
    try:
        ranklens.load_state(hip, ranklens.generate_state(2000, 200))
        reqs = ranklens.generate_requests(6, 100, 2000, 200, seed=71)
        reqs[0]["items"][0]["fields"] = [{"name": "popularity", "value": 5.5}, {"name": "genres", "value": ["drama"]}]
        reqs[1]["fields"] = [{"name": "query", "value": "socks"}, {"name": "flags", "value": [1.0, 2.0]}, {"name": "b", "value": True}]
        reqs[2]["user"] = None
        sample = np.concatenate([hip.matrix(ev) for ev in reqs])
        inner = synth.synthetic_lgbm_model(n_trees=120, n_features=24, quantiles=ranklens.column_quantiles(sample))
        names = cfg["models"]["xgboost"]["features"]
        blob = synth.write_container(names, 0, inner, warmup=[codec.ranking_event(ev) for ev in reqs[:3]])
        booster = hip.M.HipBooster.from_container(blob, names, hip.ctx)
        # the oracle with the same state and the container's inner model: what the binary path must reproduce
        from backends import OracleBackend

        oracle = OracleBackend(cfg, "xgboost")
        ranklens.load_state(oracle, ranklens.generate_state(2000, 200))
        oracle.load_model(inner, 0)
        for ev in reqs:
            _, s1, o1 = hip.ranker.rerank("xgboost", ev, booster)
            s2, o2 = hip.ranker.rerank_binary("xgboost", codec.ranking_event(ev), booster)
            assert np.array_equal(s1, s2) and o1.tolist() == o2.tolist()
            _, s3, o3 = oracle.rerank(ev)
            assert np.array_equal(s2, s3) and o2.tolist() == o3.tolist()
        assert hip.ranker.warmup("xgboost", booster) == 3
        with pytest.raises(hip.M.MrkError):
            hip.ranker.rerank_binary("xgboost", codec.ranking_event(reqs[0]), booster, capacity=10)  # more items than room
    finally:
        hip.close()
