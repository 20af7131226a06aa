"""Micro-benchmark of the text-encoder leg (all-MiniLM-L6-v2's shape, random weights -- there is no network for
checkpoints and the arithmetic does not depend on the values): single-query latency (the /rank case), batched query
throughput, and a cross-encoder batch (the only MFMA-bound work on the path).  FLOPs = 2 * tokens * matrix parameters
(+ attention 4 * seq * hidden per token per layer), padding tokens included because the kernels compute them.
  python tools/encoder_bench.py [--json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workloads import synth  # noqa: E402
from metarank_amd.encoder import HipEncoder, HipTokenizer  # noqa: E402


def flops(n, seq, layers=6, H=384, I=1536):
    per_tok = layers * (2 * (4 * H * H + 2 * H * I) + 4 * seq * H)
    return n * seq * per_tok


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return np.array(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--precision", choices=["f16", "f32"], default="f16")
    ap.add_argument("--quick", action="store_true", help="single query + the c5 batch (3 840 queries) only")
    a = ap.parse_args()
    w = synth.synthetic_bert(classifier=True)
    tj = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=128)
    enc = HipEncoder(synth.bert_safetensors(w, 12), tj, precision=a.precision)
    tok = HipTokenizer(tj)
    out = {}
    q1 = synth.synthetic_queries(1, seed=1)
    ids, ty, m = tok.encode_batch(q1)
    t = timeit(lambda: enc.embed(q1), 200)
    out["single_query"] = {"tokens": int(m.sum()), "p50_ms": float(np.percentile(t, 50) * 1e3), "p99_ms": float(np.percentile(t, 99) * 1e3)}
    t = timeit(lambda: enc.embed_ids(ids, ty, m), 200)
    out["single_query_ids"] = {"p50_ms": float(np.percentile(t, 50) * 1e3)}
    qs = synth.synthetic_queries(3840, seed=3)   # bench.py --workload c5: one device batch's queries, packed
    ids, ty, m = tok.encode_batch(qs)
    t = timeit(lambda: enc.embed_ids(ids, ty, m), 10)
    ms = float(np.median(t) * 1e3)
    real = int(m.sum())
    out["c5_batch_3840"] = {"tokens": real, "ms": ms, "tflops_real_tokens": sum(flops(1, int(l)) for l in m.sum(axis=1)) / ms / 1e9}
    for n in (() if a.quick else (64, 256, 1024)):
        qs = synth.synthetic_queries(n, seed=2)
        ids, ty, m = tok.encode_batch(qs)
        t = timeit(lambda: enc.embed_ids(ids, ty, m), 20)
        ms = float(np.median(t) * 1e3)
        out[f"queries_{n}"] = {"seq": int(ids.shape[1]), "ms": ms, "queries_per_s": n / ms * 1e3, "tflops": flops(n, ids.shape[1]) / ms / 1e9}
    for n, seq in (() if a.quick else ((100, 64), (100, 128), (1000, 64), (4096, 64))):
        rng = np.random.default_rng(1)
        ids = rng.integers(5, 2000, size=(n, seq)).astype(np.int32); ty = np.zeros_like(ids); m = np.ones_like(ids)
        t = timeit(lambda: enc.score_ids(ids, ty, m), 10)
        ms = float(np.median(t) * 1e3)
        out[f"cross_{n}x{seq}"] = {"ms": ms, "pairs_per_s": n / ms * 1e3, "tflops": flops(n, seq) / ms / 1e9}
    if a.json:
        print(json.dumps(out))
    else:
        for k, v in out.items():
            print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()})


if __name__ == "__main__":
    main()
