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

from metarank_amd import _native as N, ranklens, synth
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
    enc = HipEncoder(open(os.path.join(GOLDEN, fname), "rb").read(), TOK_TINY)
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
    enc = HipEncoder(open(os.path.join(GOLDEN, "cross_tiny.onnx"), "rb").read(), TOK_TINY)
    assert enc.info["has_classifier"] == 1
    got = enc.score_ids(g["pair_ids"], g["pair_type_ids"], g["pair_mask"])
    np.testing.assert_allclose(got, g["logits"], rtol=0, atol=ATOL_FP32)   # transformers fp32
    import json
    cases = json.load(open(os.path.join(GOLDEN, "tokenizer_cases.json"), encoding="utf-8"))
    pairs = cases["pairs"]
    np.testing.assert_array_equal(enc.score_pairs([p[0] for p in pairs], [p[1] for p in pairs]), got)  # text path == id path
    enc.close()


def test_texts_equal_ids_and_padding_does_not_change_a_row():
    enc = HipEncoder(open(os.path.join(GOLDEN, "encoder_tiny.safetensors"), "rb").read(), TOK_TINY)
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
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj)
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


def test_head_size_64(minilm):
    """BERT-base style heads (64 wide): 2 layers x 128, 2 heads"""
    w = synth.synthetic_bert(layers=2, hidden=128, heads=2, inter=256, vocab=300, max_pos=64, seed=4)
    enc = HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=300, max_length=64))
    rng = np.random.default_rng(9)
    ids = rng.integers(5, 300, size=(4, 40)); types = np.zeros_like(ids)
    mask = (np.arange(40)[None, :] < np.array([40, 3, 33, 17])[:, None]).astype(np.int32)
    h = enc.hidden_ids(ids, types, mask)
    live = mask.astype(bool)
    np.testing.assert_allclose(h[live], bert.last_hidden_state(w, ids, types, mask, heads=2, fp16=True)[live], rtol=0, atol=ATOL_MODEL)
    enc.close()


def test_load_errors():
    w = synth.synthetic_bert(layers=1, hidden=96, heads=2, inter=128, vocab=50, max_pos=16)  # head size 48
    with pytest.raises(N.MrkError) as e:
        HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=100))
    assert e.value.status == N.ERR_UNSUPPORTED
    del w["encoder.layer.0.output.dense.bias"]
    with pytest.raises(N.MrkError) as e:
        HipEncoder(synth.bert_safetensors(w, 2), synth.wordpiece_tokenizer_json(vocab_size=100))
    assert e.value.status == N.ERR_PARSE
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    enc = HipEncoder(open(os.path.join(GOLDEN, "encoder_tiny.safetensors"), "rb").read(), TOK_TINY)
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
