"""How the 128 x 128 f32 product (gemm_f32_mfma_kernel<4,4,*>) fills the chip: hidden-state calls whose token count M puts exactly
r rounds of workgroups on the 512 slots (2 per CU); run under rocprofv3 --kernel-trace and read with tools/trace_by_grid.py.
  python tools/gemm_rounds.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workloads import synth  # noqa: E402
from metarank_amd.encoder import HipEncoder  # noqa: E402

w = synth.synthetic_bert(classifier=False)
enc = HipEncoder(synth.bert_safetensors(w, 12), synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=128), precision="f32")
rng = np.random.default_rng(0)
# QKV has 9 column tiles, FFN-1 12, out-proj / FFN-2 3: m_tiles = M / 128
for m_tiles in (56, 112, 168, 42, 85, 170, 256, 258):
    M = m_tiles * 128
    n, seq = M // 64, 64
    ids = rng.integers(5, 2000, size=(n, seq)).astype(np.int32)
    mask = np.ones_like(ids)
    for _ in range(3):
        enc.hidden_ids(ids, None, mask)
    print("m_tiles", m_tiles, "M", M, flush=True)
enc.close()
