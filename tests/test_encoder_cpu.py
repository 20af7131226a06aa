"""CPU side of the text-encoder leg: the oracle (oracle/bert.py) against transformers' BERT, and the checkpoint
readers (ONNX / safetensors, host code behind mrk_encoder_load) against the safetensors library.

Fixtures: tests/golden/{encoder_tiny.onnx,encoder_tiny.safetensors,cross_tiny.onnx,encoder_tiny.npz} are made by
tools/make_encoder_golden.py from transformers.BertModel / BertForSequenceClassification (fp32) and torch.onnx.export.
"""
import os

import numpy as np
import pytest

from metarank_amd.encoder import describe_checkpoint
from oracle import bert

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _state():
    from safetensors.numpy import load_file
    return bert.strip_prefix(load_file(os.path.join(GOLDEN, "encoder_tiny.safetensors")))


def test_oracle_hidden_states_match_transformers_fixture():
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    w = _state()
    h = bert.last_hidden_state(w, g["ids"], g["type_ids"], g["mask"], heads=2)
    live = g["mask"].astype(bool)
    np.testing.assert_allclose(h[live], g["hidden"][live], rtol=0, atol=2e-5)
    # padded positions are computed too (the graph has no notion of padding beyond the additive mask)
    np.testing.assert_allclose(h, g["hidden"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(bert.avgpool(h, g["mask"]), g["pooled"], rtol=0, atol=2e-5)


def test_avgpool_known_answer():
    # OnnxBiEncoder.avgpool: mean over the first sum(mask) tokens, f64 accumulation, f32 result
    hidden = np.array([[[1, 2], [3, 4], [100, 100]], [[5, 6], [7, 8], [9, 10]]], dtype=np.float32)
    out = bert.avgpool(hidden, np.array([[1, 1, 0], [1, 1, 1]]))
    np.testing.assert_array_equal(out, np.array([[2, 3], [7, 8]], dtype=np.float32))


def test_oracle_matches_live_transformers():
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    cfg = transformers.BertConfig(vocab_size=97, hidden_size=128, num_hidden_layers=3, num_attention_heads=4, intermediate_size=256,
                                  max_position_embeddings=40, num_labels=1)
    torch.manual_seed(3)
    m = transformers.BertForSequenceClassification(cfg).eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            p.copy_(torch.randn(p.shape) * (0.2 if p.dim() > 1 else 0.1) + (1.0 if "LayerNorm.weight" in n_ else 0.0))
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 97, size=(5, 33)); tt = rng.integers(0, 2, size=(5, 33))
    lens = [33, 1, 17, 32, 8]
    mask = np.array([[1] * l + [0] * (33 - l) for l in lens])
    with torch.no_grad():
        out = m(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask), token_type_ids=torch.tensor(tt), output_hidden_states=True)
    w = bert.strip_prefix({k: v.numpy() for k, v in m.state_dict().items()})
    h = bert.last_hidden_state(w, ids, tt, mask, heads=4)
    np.testing.assert_allclose(h, out.hidden_states[-1].numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(bert.cross_logits(w, ids, tt, mask, heads=4), out.logits[:, 0].numpy(), rtol=0, atol=5e-5)


def test_cross_logits_fixture():
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    d = describe_checkpoint(open(os.path.join(GOLDEN, "cross_tiny.onnx"), "rb").read())
    assert "pooler.dense.weight" in d["tensors"] and d["tensors"]["classifier.weight"]["shape"] == [1, 64]
    assert g["logits"].shape == (g["pair_ids"].shape[0],)


@pytest.mark.parametrize("fname", ["encoder_tiny.onnx", "encoder_tiny.safetensors"])
def test_checkpoint_readers_recover_the_state_dict(fname):
    w = _state()
    d = describe_checkpoint(open(os.path.join(GOLDEN, fname), "rb").read())
    assert d["heads"] == 2  # ONNX: from the reshape of the projections; safetensors: metadata
    got = d["tensors"]
    for name, arr in w.items():
        assert name in got, name
        assert got[name]["shape"] == list(arr.shape), name   # Linear weights come back as [out, in]
        assert abs(got[name]["sum"] - float(arr.astype(np.float64).sum())) < 1e-3, name
        assert abs(got[name]["abs_sum"] - float(np.abs(arr.astype(np.float64)).sum())) < 1e-3, name
    # orientation, not just content: a transposed matrix has the same sums; check the rectangular ones by shape above
    assert got["encoder.layer.0.intermediate.dense.weight"]["shape"] == [128, 64]
    assert got["encoder.layer.0.output.dense.weight"]["shape"] == [64, 128]


def test_checkpoint_reader_rejects_garbage():
    from metarank_amd import _native as N
    with pytest.raises(N.MrkError):
        describe_checkpoint(b"\x00" * 64 + b"garbage that is neither protobuf nor safetensors" * 4)
