"""CPU restatement of the text-encoder leg of the /rank path -- TEST INFRASTRUCTURE ONLY.

The reference runs a BERT-family ONNX graph through onnxruntime and post-processes its output:
  * OnnxBiEncoder.embed / avgpool  (ml/onnx/sbert/OnnxBiEncoder.scala:13-60): output 0 of the graph is
    last_hidden_state [batch, seq, dim]; the sentence embedding is the mean over the first
    sum(attention_mask) tokens, accumulated in f64 and narrowed to f32 (no L2 normalisation);
  * OnnxCrossEncoder.encode        (ml/onnx/sbert/OnnxCrossEncoder.scala:22-51): output 0 is the
    logits [batch, 1]; the score is logits[:, 0].
The graph itself is a third-party artefact (HuggingFace `metarank/all-MiniLM-L6-v2`, `pytorch_model.onnx`, an LFS
pointer absent from the reference checkout) and onnxruntime is not installed here; what is restated below is the
published BERT architecture those files export (Devlin et al. 2018; `transformers.models.bert.modeling_bert`):
post-LN transformer, erf-GELU, additive attention mask, learned absolute positions.  Pinning: tests/test_encoder_cpu.py
checks this file against `transformers.BertModel` / `BertForSequenceClassification` (random-init, fp32) and against
the committed fixtures tests/golden/encoder_*.npz made by tools/make_encoder_golden.py from the same classes.

Everything is numpy f32 unless stated; `weights` maps HuggingFace state_dict names (without a leading "bert.") to arrays.
"""
from __future__ import annotations

import math

import numpy as np


def strip_prefix(weights: dict) -> dict:
    """BertForSequenceClassification prefixes the encoder with 'bert.'; drop it and non-tensor buffers."""
    out = {}
    for k, v in weights.items():
        if k.endswith("position_ids"):
            continue
        out[k[5:] if k.startswith("bert.") else k] = np.asarray(v, dtype=np.float32)
    return out


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps))).astype(np.float32) * g + b


_erf = np.vectorize(math.erf, otypes=[np.float64])


def gelu(x):
    return (0.5 * x.astype(np.float64) * (1.0 + _erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32)


def n_layers(w: dict) -> int:
    n = 0
    while f"encoder.layer.{n}.attention.self.query.weight" in w:
        n += 1
    return n


def _h(a):
    """round to fp16 and back: where the device path stores a value in fp16"""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def last_hidden_state(w: dict, ids, type_ids, mask, heads: int, eps: float = 1e-12, fp16: bool = False):
    """ids/type_ids/mask: int arrays [B, S] -> f32 [B, S, H]  (BertModel.forward, eval mode).

    fp16=True rounds at the points where encoder.hip keeps fp16 (matrix weights, embedding tables, the operands of every
    matrix product) and leaves everything else in f32 -- a model of the device arithmetic that separates "the kernels
    compute the right thing" (tight tolerance against this) from "fp16 is accurate enough" (against fp16=False)."""
    ids = np.asarray(ids); type_ids = np.asarray(type_ids); mask = np.asarray(mask)
    B, S = ids.shape
    r = _h if fp16 else (lambda a: a)
    if fp16:
        w = {k: (_h(v) if (v.ndim == 2 and not k.startswith("classifier")) else v) for k, v in w.items()}
    x = (w["embeddings.word_embeddings.weight"][ids] + w["embeddings.position_embeddings.weight"][np.arange(S)][None]) + \
        w["embeddings.token_type_embeddings.weight"][type_ids]
    x = layer_norm(x.astype(np.float32), w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], eps)
    H = x.shape[-1]
    dh = H // heads
    neg = np.where(mask[:, None, None, :] > 0, np.float32(0), np.finfo(np.float32).min).astype(np.float32)
    for l in range(n_layers(w)):
        p = f"encoder.layer.{l}."
        def lin(name, t):
            return t @ w[p + name + ".weight"].T + w[p + name + ".bias"]
        xh = r(x)
        q = r(lin("attention.self.query", xh)).reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
        k = r(lin("attention.self.key", xh)).reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
        v = r(lin("attention.self.value", xh)).reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / np.float32(math.sqrt(dh)) + neg
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        ctx = r((r(e) @ v) / e.sum(-1, keepdims=True)).transpose(0, 2, 1, 3).reshape(B, S, H)
        x = layer_norm(lin("attention.output.dense", ctx) + x, w[p + "attention.output.LayerNorm.weight"],
                       w[p + "attention.output.LayerNorm.bias"], eps)
        h = r(gelu(lin("intermediate.dense", r(x))))
        x = layer_norm(lin("output.dense", h) + x, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], eps)
    return x.astype(np.float32)


def avgpool(hidden, mask):
    """OnnxBiEncoder.avgpool (OnnxBiEncoder.scala:36-60): f64 sum over the first sum(mask) tokens, / count, -> f32."""
    hidden = np.asarray(hidden, dtype=np.float32)
    out = np.zeros((hidden.shape[0], hidden.shape[2]), dtype=np.float32)
    for s in range(hidden.shape[0]):
        n = int(np.asarray(mask[s]).sum())
        acc = np.zeros(hidden.shape[2], dtype=np.float64)
        for j in range(n):  # sequential f64 adds, token order
            acc += hidden[s, j].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[s] = (acc / n).astype(np.float32) if n else np.full(hidden.shape[2], np.nan, dtype=np.float32)
    return out


def embed(w: dict, ids, type_ids, mask, heads: int, eps: float = 1e-12, fp16: bool = False):
    """OnnxBiEncoder.embed after tokenisation: [B, S] ints -> f32 [B, H]."""
    return avgpool(last_hidden_state(w, ids, type_ids, mask, heads, eps, fp16), mask)


def cross_logits(w: dict, ids, type_ids, mask, heads: int, eps: float = 1e-12, fp16: bool = False):
    """OnnxCrossEncoder.encode after tokenisation: BertForSequenceClassification(num_labels=1) logits[:, 0]."""
    h = last_hidden_state(w, ids, type_ids, mask, heads, eps, fp16)[:, 0]
    pw = _h(w["pooler.dense.weight"]) if fp16 else w["pooler.dense.weight"]
    pooled = np.tanh(h @ pw.T + w["pooler.dense.bias"]).astype(np.float32)
    return (pooled @ w["classifier.weight"].T + w["classifier.bias"])[:, 0].astype(np.float32)
