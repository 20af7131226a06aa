"""Text-encoder leg on the GPU, through the C ABI (mrk_encoder_*), against oracle/bert.py and the transformers fixtures.

Tolerances (floating point, stated per the tier rules):
  * against the oracle run with the device's rounding points (fp16 weights / matrix-product operands, everything else
    f32; oracle `fp16=True`): 4e-3 absolute on unit-scale hidden states -- what is left is f32 summation order, the online
    softmax's running maximum and exp/erf/tanh implementations; this is the "kernels compute the right thing" bar;
  * against the fp32 graph (transformers / oracle fp32): 3e-2 absolute on hidden states, 3e-3 on cosines -- the cost of
    fp16 storage (BASELINE config 5 asks for fp16); the reference's own encoder tests accept 1e-3 on cosines of real
    sentence embeddings (OnnxBiencoderTest.scala:23-25).
"""
import os

import numpy as np
import pytest

from metarank_amd import _native as N
from workloads import ranklens, synth
from metarank_amd.encoder import HipEncoder, HipTokenizer
from oracle import bert
from backends import HipBackend, OracleBackend

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOK_TINY = open(os.path.join(GOLDEN, "tokenizer_tiny.json"), "rb").read()
ATOL_MODEL, ATOL_FP32, ATOL_COS = 4e-3, 3e-2, 3e-3


def _cos(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return (a * b).sum(-1) / np.sqrt((a * a).sum(-1) * (b * b).sum(-1))


def _tiny_state():
    from safetensors.numpy import load_file
    return bert.strip_prefix(load_file(os.path.join(GOLDEN, "encoder_tiny.safetensors")))


@pytest.mark.parametrize("fname", ["encoder_tiny.onnx", "encoder_tiny.safetensors"])
def test_tiny_fixture_hidden_states_and_pool(fname):
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    enc = HipEncoder(open(os.path.join(GOLDEN, fname), "rb").read(), TOK_TINY, precision="f16")
    assert (enc.info["layers"], enc.info["hidden"], enc.info["heads"], enc.info["intermediate"]) == (2, 64, 2, 128)
    assert enc.info["has_classifier"] == 0 and enc.info["max_length"] == 24
    h = enc.hidden_ids(g["ids"], g["type_ids"], g["mask"])
    w = _tiny_state()
    model = bert.last_hidden_state(w, g["ids"], g["type_ids"], g["mask"], heads=2, fp16=True)
    np.testing.assert_allclose(h, model, rtol=0, atol=ATOL_MODEL)
    np.testing.assert_allclose(h, g["hidden"], rtol=0, atol=ATOL_FP32)      # transformers fp32
    pooled = enc.embed_ids(g["ids"], g["type_ids"], g["mask"])
    np.testing.assert_array_equal(pooled, bert.avgpool(h, g["mask"]))      # the pool itself is exact: f64 sums in token order
    assert np.abs(_cos(pooled, g["pooled"]) - 1.0).max() < ATOL_COS
    with pytest.raises(N.MrkError) as e:
        enc.score_ids(g["ids"], g["type_ids"], g["mask"])
    assert e.value.status == N.ERR_UNSUPPORTED
    enc.close()


def test_tiny_cross_encoder_logits():
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    enc = HipEncoder(open(os.path.join(GOLDEN, "cross_tiny.onnx"), "rb").read(), TOK_TINY, precision="f16")
    assert enc.info["has_classifier"] == 1
    got = enc.score_ids(g["pair_ids"], g["pair_type_ids"], g["pair_mask"])
    np.testing.assert_allclose(got, g["logits"], rtol=0, atol=ATOL_FP32)   # transformers fp32
    import json
    cases = json.load(open(os.path.join(GOLDEN, "tokenizer_cases.json"), encoding="utf-8"))
    pairs = cases["pairs"]
    np.testing.assert_array_equal(enc.score_pairs([p[0] for p in pairs], [p[1] for p in pairs]), got)  # text path == id path
    enc.close()


def test_texts_equal_ids_and_padding_does_not_change_a_row():
    enc = HipEncoder(open(os.path.join(GOLDEN, "encoder_tiny.safetensors"), "rb").read(), TOK_TINY, precision="f16")
    tok = HipTokenizer(TOK_TINY)
    texts = ["star wars", "the quick brown fox jumps over the lazy dog and the terminator again and again", "café", ""]
    ids, types, mask = tok.encode_batch(texts)
    a = enc.embed(texts)
    np.testing.assert_array_equal(a, enc.embed_ids(ids, types, mask))
    for i, t in enumerate(texts):  # batchEncode pads to the longest row; masked keys get probability exactly 0
        np.testing.assert_array_equal(enc.embed([t])[0], a[i])
    enc.close()


MINILM = dict(layers=6, hidden=384, heads=12, inter=1536, vocab=2000, max_pos=512)


@pytest.fixture(scope="module")
def minilm():
    w = synth.synthetic_bert(**MINILM, classifier=True)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj, precision="f16")
    yield w, enc, HipTokenizer(tj)
    enc.close()


@pytest.mark.parametrize("n,seq", [(1, 9), (5, 47), (3, 200), (70, 64)])
def test_minilm_shape_against_oracle(minilm, n, seq):
    """all-MiniLM-L6-v2's architecture (6 x 384, 12 heads of 32, FFN 1536) with random weights; (70, 64) takes the
    128x128 GEMM tiles, (3, 200) several key blocks of the online softmax, (1, 9) is the single-query case of /rank."""
    w, enc, _ = minilm
    rng = np.random.default_rng(n * 1000 + seq)
    ids = rng.integers(5, 2000, size=(n, seq)); types = np.zeros_like(ids)
    lens = rng.integers(1, seq + 1, size=n); lens[0] = seq
    mask = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int32)
    types[:, seq // 2:] = 1
    h = enc.hidden_ids(ids, types, mask)
    live = mask.astype(bool)
    model = bert.last_hidden_state(w, ids, types, mask, heads=12, fp16=True)
    np.testing.assert_allclose(h[live], model[live], rtol=0, atol=ATOL_MODEL)
    fp32 = bert.last_hidden_state(w, ids, types, mask, heads=12)
    np.testing.assert_allclose(h[live], fp32[live], rtol=0, atol=ATOL_FP32)
    pooled = enc.embed_ids(ids, types, mask)
    np.testing.assert_array_equal(pooled, bert.avgpool(h, mask))
    assert np.abs(_cos(pooled, bert.avgpool(fp32, mask)) - 1.0).max() < 1e-4
    logits = enc.score_ids(ids, types, mask)
    np.testing.assert_allclose(logits, bert.cross_logits(w, ids, types, mask, heads=12, fp16=True), rtol=0, atol=ATOL_MODEL)


def test_a_row_does_not_depend_on_the_batch_it_travels_in(minilm):
    """The three GEMM kernels (skinny: M <= 64 rows; 64x64 tiles; 128x128 tiles) chain their MFMAs in the same k order
    and share one epilogue, so a sequence's hidden states are the same bits whatever else is in the batch -- the
    EmbeddingCache can hold an embedding computed in any batch."""
    w, enc, _ = minilm
    rng = np.random.default_rng(77)
    seq = 20
    ids = rng.integers(5, 2000, size=(1700, seq)).astype(np.int32)
    mask = np.ones_like(ids); mask[1::2, 13:] = 0
    alone = enc.hidden_ids(ids[:1], None, mask[:1])            # M = 20: skinny kernel
    few = enc.hidden_ids(ids[:6], None, mask[:6])              # M = 120: 64x64 tiles
    many = enc.hidden_ids(ids, None, mask)                     # M = 34000: 128x128 tiles
    np.testing.assert_array_equal(alone[0], few[0])
    np.testing.assert_array_equal(few[:6], many[:6])
    import os
    os.environ["MRK_ENCODER_GRAPH"] = "1"                      # recorded graph replay vs direct launches
    N.reload_switches()
    try:
        np.testing.assert_array_equal(enc.hidden_ids(ids[:1], None, mask[:1]), alone)
    finally:
        del os.environ["MRK_ENCODER_GRAPH"]
        N.reload_switches()


def test_packed_batches_give_the_padded_bits(minilm):
    """Pooled embeddings and pair logits run over PACKED tokens (no padding: csrc/capi_encoder.cpp run_encoder); the bits
    equal those of the padded batch (MRK_ENCODER_PACKED=0) for ragged lengths 1..seq, a batch without padding, and one
    whose mask is not a prefix (which stays padded)."""
    import os
    w, enc, _ = minilm
    rng = np.random.default_rng(5)
    n, seq = 333, 29
    ids = rng.integers(5, 2000, size=(n, seq)).astype(np.int32)
    types = (np.arange(seq)[None, :] >= rng.integers(1, seq, size=(n, 1))).astype(np.int32)
    lens = rng.integers(1, seq + 1, size=n); lens[:3] = [1, seq, 2]
    mask = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int32)
    got = {}
    for packed in ("1", "0"):
        os.environ["MRK_ENCODER_PACKED"] = packed
        N.reload_switches()
        try:
            got[packed] = (enc.embed_ids(ids, types, mask), enc.score_ids(ids, types, mask),
                           enc.embed_ids(ids[:, :7], types[:, :7], np.ones((n, 7), dtype=np.int32)))
        finally:
            del os.environ["MRK_ENCODER_PACKED"]
            N.reload_switches()
    for a, b in zip(got["1"], got["0"]):
        np.testing.assert_array_equal(a, b)
    holes = mask.copy(); holes[5, :4] = [1, 0, 1, 1]   # row 5's mask is not a prefix of ones: the whole batch stays padded
    e = enc.embed_ids(ids, types, holes)
    keep = np.arange(n) != 5
    np.testing.assert_array_equal(e[keep], got["0"][0][keep])
    assert np.isfinite(e[5]).all()


def test_head_size_64(minilm):
    """BERT-base style heads (64 wide): 2 layers x 128, 2 heads"""
    w = synth.synthetic_bert(layers=2, hidden=128, heads=2, inter=256, vocab=300, max_pos=64, seed=4)
    enc = HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=300, max_length=64), precision="f16")
    rng = np.random.default_rng(9)
    ids = rng.integers(5, 300, size=(4, 40)); types = np.zeros_like(ids)
    mask = (np.arange(40)[None, :] < np.array([40, 3, 33, 17])[:, None]).astype(np.int32)
    h = enc.hidden_ids(ids, types, mask)
    live = mask.astype(bool)
    np.testing.assert_allclose(h[live], bert.last_hidden_state(w, ids, types, mask, heads=2, fp16=True)[live], rtol=0, atol=ATOL_MODEL)
    enc.close()


def test_f32_attention_with_dead_key_blocks_before_the_first_live_key():
    """Masks that are not a prefix of ones: whole 16-key blocks of dead keys BEFORE the first live key (the online softmax
    of attention_f32_mfma_kernel gave such keys weight exp(0) while its running maximum was still -inf), in the middle and at the end.
    Live positions must match the numpy fp32 graph; the vector-unit twin must agree with the matrix-core kernel."""
    w = synth.synthetic_bert(**MINILM, classifier=False)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj, precision="f32")
    try:
        rng = np.random.default_rng(77)
        n, seq = 5, 80
        ids = rng.integers(5, 2000, size=(n, seq)); types = np.zeros_like(ids)
        mask = np.zeros((n, seq), dtype=np.int32)
        mask[0, 40:52] = 1                 # two dead blocks, then live keys, then dead to the end
        mask[1, 16:] = 1                   # exactly one dead block first
        mask[2, :20] = 1; mask[2, 64:70] = 1   # dead blocks in the middle
        mask[3, 79] = 1                    # one live key, the last
        mask[4, :] = 1
        fp32 = bert.last_hidden_state(w, ids, types, mask, heads=12)
        h = enc.hidden_ids(ids, types, mask)
        live = mask.astype(bool)
        assert np.isfinite(h[live]).all()
        np.testing.assert_allclose(h[live], fp32[live], rtol=0, atol=ATOL_F32_HIDDEN)
    finally:
        enc.close()


def test_load_errors():
    w = synth.synthetic_bert(layers=1, hidden=96, heads=2, inter=128, vocab=50, max_pos=16)  # head size 48
    with pytest.raises(N.MrkError) as e:
        HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=100))
    assert e.value.status == N.ERR_UNSUPPORTED
    w = synth.synthetic_bert(layers=1, hidden=64, heads=2, inter=128, vocab=50, max_pos=16)
    del w["encoder.layer.0.output.dense.bias"]
    with pytest.raises(N.MrkError) as e:
        HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=100))
    assert e.value.status == N.ERR_PARSE
    with pytest.raises(N.MrkError) as e:  # heads unknown: no metadata, none passed
        from safetensors.numpy import save
        HipEncoder(save(synth.synthetic_bert(layers=1, hidden=64, heads=2, inter=128, vocab=50, max_pos=16)), synth.wordpiece_tokenizer_json(vocab_size=100))
    assert e.value.status == N.ERR_INVALID_ARG
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    enc = HipEncoder(open(os.path.join(GOLDEN, "encoder_tiny.safetensors"), "rb").read(), TOK_TINY, precision="f16")
    with pytest.raises(N.MrkError) as e:  # 48 positions in the tiny model
        enc.hidden_ids(np.zeros((1, 49), np.int32), None, np.ones((1, 49), np.int32))
    assert e.value.status == N.ERR_INVALID_ARG
    enc.close()


N_ITEMS, N_SESS = 3000, 300


def test_c5_query_encoded_on_the_device(minilm):
    """BASELINE config 5 end to end: the request carries the query TEXT; mrk_rank embeds it with the bound encoder
    (tokenise -> 6-layer forward -> masked mean pool), takes the cosine against the stored item embeddings, scores and
    sorts.  Checked against the assembly oracle fed the same embedding (bit-exact matrix / scores / order) and against
    the fp32 oracle embedding (cosine column within ATOL_COS)."""
    w, enc, tok = minilm
    cfg = ranklens.c5_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for b in (orc, hip):
            ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
            ranklens.load_state(b, ranklens.c5_embeddings(N_ITEMS))
        hip.ranker.bind_encoder("title_match", enc)
        queries = synth.synthetic_queries(9, seed=12)
        reqs = ranklens.generate_requests(12, 100, N_ITEMS, N_SESS, seed=52)
        text_reqs, emb_reqs = [], []
        dev_emb = enc.embed(queries)
        ids, types, mask = tok.encode_batch(queries)
        fp32_emb = bert.embed(w, ids, types, mask, heads=12)
        for k, ev in enumerate(reqs):
            t, e = dict(ev), dict(ev)
            if k % 4 != 3:  # every fourth request has no query: NaN column
                q = k % len(queries)
                if k % 2:
                    t["fields"] = [{"name": "query", "value": queries[q]}]
                else:  # StringListField: joined with " " (FieldMatchBiencoderFeature.scala:91)
                    t["fields"] = [{"name": "query", "value": queries[q].split(" ")}]
                e["fields"] = [{"name": "__embedding:title_match", "value": [float(x) for x in dev_emb[q]]}]
            text_reqs.append(t); emb_reqs.append(e)
        mats = [orc.matrix(ev) for ev in emb_reqs]
        blob = synth.synthetic_lgbm_model(n_trees=500, n_features=25, quantiles=ranklens.column_quantiles(np.concatenate(mats)),
                                          cat_features=[7], cat_prob=0.01, missing="per_feature")
        orc.load_model(blob, 0)
        hip.load_model(blob, 0)
        batch = hip.ranker.prepare("xgboost", text_reqs)
        batch.run(hip.booster)
        scores, order, mat = batch.fetch(matrix=True)
        assert (batch.status() == 0).all()
        def same(a, b):
            a, b = np.asarray(a), np.asarray(b)
            return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
        for r, ev in enumerate(emb_reqs):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            _, es, eo = orc.rerank(ev)
            assert same(mat[lo:hi], mats[r]), r
            assert same(scores[lo:hi], es), r
            assert order[lo:hi].tolist() == eo.tolist(), r
        batch.close()
        # the single-request entry point takes the same route
        m1, s1, o1 = hip.ranker.rerank("xgboost", text_reqs[1], hip.booster, explain=True)
        assert same(m1, mats[1])
        # cosine column against the fp32 graph's embedding
        k = 1
        e32 = dict(reqs[k]); e32["fields"] = [{"name": "__embedding:title_match", "value": [float(x) for x in fp32_emb[k % len(queries)]]}]
        col32 = orc.matrix(e32)[:, 24]
        ok = np.isfinite(col32)
        assert ok.sum() > 50 and np.abs(col32[ok] - mats[k][:, 24][ok]).max() < ATOL_COS
    finally:
        hip.close()


def test_cross_encoder_column(minilm):
    """FieldMatchCrossEncoderFeature end to end: item texts are stored (Put SString under the item scope), the request
    carries the query text, the device scores every (query, item text) pair; items without a text are NaN, requests
    without a query are NaN, a score the caller supplies ("__ext:<name>", the ScoreCache hit) wins; `norm` (noop | linear
    | position) is applied on the device over the request's whole column.  The matrix, scores
    and order must equal the assembly oracle's when it is handed the same logits; the logits themselves are checked
    against oracle/bert.py."""
    w, enc, tok = minilm
    def scale(vals, norm):  # ml/onnx/Normalize.scala:13-45 over one request's column (NaN = missing)
        v = np.array(vals, dtype=np.float64)
        ok = ~np.isnan(v)
        if norm == "linear" and ok.any():
            v = (v - v[ok].min()) / (v[ok].max() - v[ok].min())
        elif norm == "position":
            order = sorted(range(len(v)), key=lambda i: (bool(np.isnan(v[i])), 0.0 if np.isnan(v[i]) else v[i], i))  # stable, NaN last
            out = v.copy()
            for s_, i in enumerate(order):
                if ok[i]:
                    out[i] = s_ / len(v)
            v = out
        return v

    for norm in ("noop", "linear", "position"):
        cfg = ranklens.ranklens_config()
        cfg["features"].append({"name": "title_cross", "type": "field_match", "itemField": "item.title", "rankingField": "ranking.query",
                                "method": {"type": "cross-encoder", "model": "metarank/ce-msmarco-MiniLM-L6-v2"}, "norm": norm})
        cfg["models"]["xgboost"]["features"] = cfg["models"]["xgboost"]["features"] + ["title_cross"]
        orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
        try:
            for b in (orc, hip):
                ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
            titles = synth.synthetic_queries(N_ITEMS, seed=21, words=(1, 7))
            rng = np.random.default_rng(8)
            has_text = rng.random(N_ITEMS) > 0.05
            for i in range(N_ITEMS):
                if has_text[i]:
                    hip.ranker.put_string(f"item={i}/title_cross", titles[i])
            hip.ranker.put_string("item=0/title_cross", "to be deleted")
            hip.ranker.delete("item=0/title_cross")
            has_text[0] = False
            hip.ranker.bind_encoder("title_cross", enc)
            queries = synth.synthetic_queries(6, seed=31)
            reqs = ranklens.generate_requests(6, 60, N_ITEMS, N_SESS, seed=53)
            text_reqs, ext_reqs, all_logits = [], [], []
            for k, ev in enumerate(reqs):
                t, e = dict(ev), dict(ev)
                if k != 4:  # request 4 has no query: NaN column
                    t["fields"] = [{"name": "query", "value": queries[k]}]
                    known = [it["id"] for it in ev["items"] if it["id"].isdigit() and has_text[int(it["id"])]]
                    logits = enc.score_pairs([queries[k]] * len(known), [titles[int(i)] for i in known])
                    all_logits.append((queries[k], [titles[int(i)] for i in known], logits))
                    raw = dict(zip(known, logits.astype(np.float64)))
                    if k == 2:  # ScoreCache hit for the first item: the given raw score is used, the pair is not encoded
                        first = ev["items"][0]["id"]
                        raw[first] = 0.125
                        t["items"] = [dict(it, fields=[{"name": "__ext:title_cross", "value": 0.125}]) if it["id"] == first else it for it in ev["items"]]
                    # schema.norm.scale over the request's whole column: encoded logits + given scores, NaN for the rest
                    vals = scale([raw.get(it["id"], np.nan) for it in ev["items"]], norm)
                    e["items"] = [dict(it, fields=[{"name": "__ext:title_cross", "value": float(v)}]) if v == v else it
                                  for it, v in zip(ev["items"], vals)]
                text_reqs.append(t); ext_reqs.append(e)
            mats = [orc.matrix(ev) for ev in ext_reqs]
            col = np.concatenate(mats)[:, 24]
            assert np.isnan(col).any() and np.isfinite(col).sum() > 200
            blob = synth.synthetic_lgbm_model(n_trees=200, n_features=25, quantiles=ranklens.column_quantiles(np.concatenate(mats)),
                                              cat_features=[7], cat_prob=0.01, missing="per_feature")
            orc.load_model(blob, 0)
            hip.load_model(blob, 0)
            batch = hip.ranker.prepare("xgboost", text_reqs)
            batch.run(hip.booster)
            scores, order, mat = batch.fetch(matrix=True)
            assert (batch.status() == 0).all()

            def same(a, b):
                a, b = np.asarray(a), np.asarray(b)
                return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
            for r, ev in enumerate(ext_reqs):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                _, es, eo = orc.rerank(ev)
                assert same(mat[lo:hi], mats[r]), (norm, r)
                assert same(scores[lo:hi], es), (norm, r)
                assert order[lo:hi].tolist() == eo.tolist(), (norm, r)
            batch.close()
            if norm == "noop":  # the logits against the CPU restatement of the graph
                q, ts, logits = all_logits[0]
                ids, types, mask = tok.encode_batch([q] * len(ts), ts)
                want = bert.cross_logits(w, ids, types, mask, heads=12, fp16=True)
                np.testing.assert_allclose(logits, want, rtol=0, atol=ATOL_MODEL)
        finally:
            hip.close()


# ---- precision f32 (mrk_encoder_load_ex(MRK_ENCODER_F32)): the fp32 ONNX session's arithmetic --------------------------
ATOL_F32_HIDDEN, ATOL_F32_COS = 2e-4, 1e-5   # vs transformers / numpy fp32: what is left is f32 summation order and exp / erf


def test_f32_mode_reproduces_the_fp32_graph():
    """Tiny fixtures (transformers fp32 outputs, ONNX and safetensors readers) and all-MiniLM-L6-v2's architecture: hidden
    states within 2e-4, pooled cosines within 1e-5 of the fp32 graph - the reference's own encoder tests accept 1e-3
    (OnnxBiencoderTest.scala:23-25); the fp16 path is within 3e-3 (ATOL_COS above)."""
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    for fname in ("encoder_tiny.onnx", "encoder_tiny.safetensors"):
        enc = HipEncoder(open(os.path.join(GOLDEN, fname), "rb").read(), TOK_TINY, f32=True)
        h = enc.hidden_ids(g["ids"], g["type_ids"], g["mask"])
        live = g["mask"].astype(bool)
        np.testing.assert_allclose(h[live], g["hidden"][live], rtol=0, atol=ATOL_F32_HIDDEN)   # transformers fp32
        pooled = enc.embed_ids(g["ids"], g["type_ids"], g["mask"])
        np.testing.assert_array_equal(pooled, bert.avgpool(h, g["mask"]))
        assert np.abs(_cos(pooled, g["pooled"]) - 1.0).max() < ATOL_F32_COS
        enc.close()
    enc = HipEncoder(open(os.path.join(GOLDEN, "cross_tiny.onnx"), "rb").read(), TOK_TINY, f32=True)
    np.testing.assert_allclose(enc.score_ids(g["pair_ids"], g["pair_type_ids"], g["pair_mask"]), g["logits"], rtol=0, atol=ATOL_F32_HIDDEN)
    enc.close()
    w = synth.synthetic_bert(**MINILM, classifier=True)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj, f32=True)
    try:
        for n, seq in ((1, 9), (5, 47), (3, 200), (70, 64)):
            rng = np.random.default_rng(n * 1000 + seq)
            ids = rng.integers(5, 2000, size=(n, seq)); types = np.zeros_like(ids)
            lens = rng.integers(1, seq + 1, size=n); lens[0] = seq
            mask = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int32)
            types[:, seq // 2:] = 1
            fp32 = bert.last_hidden_state(w, ids, types, mask, heads=12)
            h = enc.hidden_ids(ids, types, mask)
            live = mask.astype(bool)
            np.testing.assert_allclose(h[live], fp32[live], rtol=0, atol=ATOL_F32_HIDDEN)
            pooled = enc.embed_ids(ids, types, mask)     # packed layout: the same rows
            assert np.abs(_cos(pooled, bert.avgpool(fp32, mask)) - 1.0).max() < ATOL_F32_COS
            np.testing.assert_allclose(pooled, bert.avgpool(fp32, mask), rtol=0, atol=ATOL_F32_HIDDEN)
            logits = enc.score_ids(ids, types, mask)
            np.testing.assert_allclose(logits, bert.cross_logits(w, ids, types, mask, heads=12), rtol=0, atol=5e-4)
    finally:
        enc.close()


def test_f32_products_on_the_matrix_cores_are_the_fma_chain_and_batch_independent():
    """The f32 mode's products and attention run on v_mfma_f32_16x16x4_f32 (exact f32: bit for bit a chain of fma's).  All
    three product kernels (<= 32 rows and 64 x 64 tiles on 16x16x4; 128 x 128 tiles on 32x32x2) walk k in one canonical order from a zero accumulator,
    so (1) the vector-unit twin (MRK_ENCODER_F32_MFMA=0: v_fma_f32 in the same order) gives the same BITS for every product -
    hidden states then differ only through the two attention kernels' softmax bookkeeping, within 1e-5 -, and (2) a sequence's
    hidden states are the same bits alone (skinny kernel), among a few (64 x 64 tiles), among thousands (128 x 128 tiles), and
    (3) packed == padded."""
    import os
    w = synth.synthetic_bert(**MINILM, classifier=True)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj, precision="f32")
    try:
        rng = np.random.default_rng(78)
        seq = 20
        ids = rng.integers(5, 2000, size=(3500, seq)).astype(np.int32)
        mask = np.ones_like(ids); mask[1::2, 13:] = 0
        alone = enc.hidden_ids(ids[:1], None, mask[:1])            # M = 20: skinny kernel, MT = 2
        one9 = enc.hidden_ids(ids[:1, :9], None, mask[:1, :9])     # M = 9: skinny kernel, MT = 1
        few = enc.hidden_ids(ids[:6], None, mask[:6])              # M = 120: 64 x 64 tiles
        many = enc.hidden_ids(ids, None, mask)                     # M = 70 000: 128 x 128 tiles
        np.testing.assert_array_equal(alone[0], few[0])
        np.testing.assert_array_equal(few[:6], many[:6])
        assert np.isfinite(many).all()
        nine = enc.hidden_ids(np.repeat(ids[:1, :9], 40, axis=0), None, np.ones((40, 9), dtype=np.int32))
        np.testing.assert_array_equal(nine[7], one9[0])
        fp32 = bert.last_hidden_state(w, ids[:6], np.zeros_like(ids[:6]), mask[:6], heads=12)
        live = mask[:6].astype(bool)
        np.testing.assert_allclose(few[live], fp32[live], rtol=0, atol=ATOL_F32_HIDDEN)
        # packed == padded
        n = 333
        lens = rng.integers(1, seq + 1, size=n); lens[:3] = [1, seq, 2]
        m2 = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int32)
        got = {}
        for packed in ("1", "0"):
            os.environ["MRK_ENCODER_PACKED"] = packed
            N.reload_switches()
            try:
                got[packed] = (enc.embed_ids(ids[:n], None, m2), enc.score_ids(ids[:n], None, m2))
            finally:
                del os.environ["MRK_ENCODER_PACKED"]
                N.reload_switches()
        for a, b in zip(got["1"], got["0"]):
            np.testing.assert_array_equal(a, b)
        # the vector-unit twin: the embedding + first QKV product + ... are the same chain; what differs is attention's
        # bookkeeping (running maximum over 16-key blocks vs one pass), so compare within the f32 tolerance AND require that
        # a single layer's worth of products is bit-identical: sequences of ONE token have a softmax of exactly 1.0 in both
        os.environ["MRK_ENCODER_F32_MFMA"] = "0"
        N.reload_switches()
        try:
            valu_few = enc.hidden_ids(ids[:6], None, mask[:6])
            valu_one_tok = enc.hidden_ids(ids[:300, :1], None, np.ones((300, 1), dtype=np.int32))
        finally:
            del os.environ["MRK_ENCODER_F32_MFMA"]
            N.reload_switches()
        np.testing.assert_allclose(valu_few[live], few[live], rtol=0, atol=1e-5)
        mfma_one_tok = enc.hidden_ids(ids[:300, :1], None, np.ones((300, 1), dtype=np.int32))
        np.testing.assert_array_equal(valu_one_tok, mfma_one_tok)
    finally:
        enc.close()


def test_precision_is_a_property_of_the_handle_not_of_the_call():
    """mrk_encoder_load (and the ABI <= 7 name MRK_ENCODER_AUTO) = f32: a text's embedding is the same bits in a call of one
    (mrk_rank's query), of three, of twelve (a packed batch), as ids or as text, cached or not - so a request's scores do not
    depend on how many callers the combining front merged.  fp16 is a different handle, and different bits."""
    w = synth.synthetic_bert(**MINILM, classifier=False)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    blob = synth.bert_safetensors(w, 12)
    auto, e16, e32, dflt = HipEncoder(blob, tj, precision="auto"), HipEncoder(blob, tj, precision="f16"), HipEncoder(blob, tj, precision="f32"), HipEncoder(blob, tj)
    try:
        texts = synth.synthetic_queries(300, seed=5)
        all32 = e32.embed(texts)
        np.testing.assert_array_equal(auto.embed(texts[:3]), all32[:3])
        np.testing.assert_array_equal(auto.embed(texts), all32)
        np.testing.assert_array_equal(dflt.embed(texts[5:6]), all32[5:6])
        np.testing.assert_array_equal(dflt.embed(texts[:12]), all32[:12])
        for k in (0, 7, 299):
            np.testing.assert_array_equal(HipEncoder(blob, tj).embed(texts[k:k + 1]), all32[k:k + 1])   # a fresh handle: no cache
        assert not np.array_equal(e16.embed(texts[:3]), all32[:3])
        ids, types, mask = HipTokenizer(tj).encode_batch(texts)
        np.testing.assert_array_equal(dflt.embed_ids(ids[:4], types[:4], mask[:4]), all32[:4])
        np.testing.assert_array_equal(dflt.embed_ids(ids, types, mask), all32)
    finally:
        for e in (auto, e16, e32, dflt):
            e.close()


def test_c5_against_the_fp32_embedding_not_against_itself():
    """BASELINE config 5 with the oracle fed an INDEPENDENT embedding - numpy's fp32 run of the graph - instead of the device's
    own: with the encoder in f32 mode the device's cosine column is within 2e-6 of the oracle's, and every score the
    500-tree forest produces is the oracle's except where the two cosines STRADDLE one of the forest's split thresholds on that
    column (asserted per moved score, the count printed); the same
    comparison for the default fp16 mode is reported (how many of the scores move by more than 1e-5 at 3e-3 cosine error)."""
    w = synth.synthetic_bert(**MINILM, classifier=False)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=256)
    blob_w = synth.bert_safetensors(w, 12)
    tok = HipTokenizer(tj)
    cfg = ranklens.c5_config()
    queries = synth.synthetic_queries(24, seed=13)
    ids, types, mask = tok.encode_batch(queries)
    fp32_emb = bert.embed(w, ids, types, mask, heads=12)
    reqs = ranklens.generate_requests(24, 100, N_ITEMS, N_SESS, seed=53)
    text_reqs, emb_reqs = [], []
    for k, ev in enumerate(reqs):
        t, e = dict(ev), dict(ev)
        t["fields"] = [{"name": "query", "value": queries[k]}]
        e["fields"] = [{"name": "__embedding:title_match", "value": [float(x) for x in fp32_emb[k]]}]
        text_reqs.append(t); emb_reqs.append(e)
    orc = OracleBackend(cfg, "xgboost")
    ranklens.load_state(orc, ranklens.generate_state(N_ITEMS, N_SESS))
    ranklens.load_state(orc, ranklens.c5_embeddings(N_ITEMS))
    mats = [orc.matrix(ev) for ev in emb_reqs]
    model = synth.synthetic_lgbm_model(n_trees=500, n_features=25, quantiles=ranklens.column_quantiles(np.concatenate(mats)), missing="per_feature")
    orc.load_model(model, 0)
    want = [orc.rerank(ev) for ev in emb_reqs]
    # the forest's split thresholds on the cosine column (24), read from the model text itself
    cos_thr = []
    feats = None
    for line in model.decode().split("\n"):
        if line.startswith("split_feature="):
            feats = [int(x) for x in line.split("=", 1)[1].split()]
        elif line.startswith("threshold=") and feats is not None:
            cos_thr += [float(t) for f, t in zip(feats, line.split("=", 1)[1].split()) if f == 24]
            feats = None
    cos_thr = np.unique(np.array(cos_thr))
    assert len(cos_thr) > 20
    straddled = []
    report = {}
    for mode in ("f32", "fp16"):
        enc = HipEncoder(blob_w, tj, precision="f32" if mode == "f32" else "f16")
        hip = HipBackend(cfg, "xgboost")
        try:
            ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
            ranklens.load_state(hip, ranklens.c5_embeddings(N_ITEMS))
            hip.ranker.bind_encoder("title_match", enc)
            hip.load_model(model, 0)
            batch = hip.ranker.prepare("xgboost", text_reqs)
            batch.run(hip.booster)
            scores, order, mat = batch.fetch(matrix=True)
            assert (batch.status() == 0).all()
            cos_err, moved, total, reordered = 0.0, 0, 0, 0
            for r in range(len(reqs)):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                a, b = mat[lo:hi], mats[r]
                other = [c for c in range(25) if c != 24]
                assert bool(((a[:, other] == b[:, other]) | (np.isnan(a[:, other]) & np.isnan(b[:, other]))).all()), (mode, r)
                ok = np.isfinite(b[:, 24])
                cos_err = max(cos_err, float(np.abs(a[ok, 24] - b[ok, 24]).max()))
                mv = np.flatnonzero(np.abs(scores[lo:hi] - want[r][1]) > 1e-5)
                moved += len(mv)
                if mode == "f32":   # every moved score is EXPLAINED: the two cosines of that candidate straddle a split threshold of the column
                    for i in mv:
                        x, y = sorted((float(a[i, 24]), float(b[i, 24])))
                        k = np.searchsorted(cos_thr, x, side="left")   # LightGBM: v <= threshold goes left
                        assert k < len(cos_thr) and x <= cos_thr[k] < y, (r, int(i), x, y)
                        assert y - x < 2e-6, (r, int(i), x, y)
                        straddled.append(y - x)
                total += hi - lo
                reordered += int(order[lo:hi].tolist() != want[r][2].tolist())
            report[mode] = {"cosine_err": cos_err, "scores_moved": moved, "of": total, "requests_reordered": reordered}
            if mode == "f32":   # the same requests ONE AT A TIME (mrk_rank: the <= 32-row product kernel, a fresh handle - no cached
                # embedding): the bits of the packed batch - north_star's "identical ordering" between the two entry points by construction
                enc1 = HipEncoder(blob_w, tj)
                hip.ranker.bind_encoder("title_match", enc1)
                for r, ev in enumerate(text_reqs):
                    lo, hi = batch.offsets[r], batch.offsets[r + 1]
                    _, s1, o1 = hip.rerank(ev)
                    np.testing.assert_array_equal(s1, scores[lo:hi])
                    assert o1.tolist() == order[lo:hi].tolist(), r
                hip.ranker.bind_encoder("title_match", enc)
                enc1.close()
            batch.close()
        finally:
            hip.close()
            enc.close()
    report["f32"]["straddled_thresholds"] = len(straddled)
    report["f32"]["widest_straddle"] = max(straddled) if straddled else 0.0
    print("\nC5 against the fp32 embedding:", report)
    # the tolerance is explained, not blanket: each of the f32 mode's moved scores (printed above) belongs to a candidate whose
    # cosine - 1e-6 apart between two f32 orderings of the same graph - has one of the forest's thresholds between the two values
    assert report["f32"]["cosine_err"] < 2e-6 and report["f32"]["scores_moved"] <= report["f32"]["of"] // 200, report
    assert report["fp16"]["cosine_err"] < ATOL_COS, report
