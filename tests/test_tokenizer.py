"""WordPiece tokenizer (host side of the text-encoder leg) against the HuggingFace `tokenizers` library -- the library
DJL's HuggingFaceTokenizer wraps (OnnxSession.scala:42-43) -- through the C ABI (mrk_tokenizer_*).

  * tests/golden/tokenizer_cases.json: outputs of the library for fixed texts / pairs under six pipeline variants
    (made by tools/make_encoder_golden.py); always run;
  * a live randomised comparison with the library when it is importable (it is in the image).
"""
import json
import os

import numpy as np
import pytest

from metarank_amd import _native as N
from metarank_amd.encoder import HipTokenizer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLDEN, "tokenizer_cases.json"), encoding="utf-8"))


def _json_of(variant):
    v = CASES["variants"][variant]
    if "json" in v:
        return open(os.path.join(GOLDEN, v["json"]), "rb").read()
    return v["json_text"].encode("utf-8")


def _rows(ids, types, mask):
    return [{"ids": list(map(int, i)), "type_ids": list(map(int, t)), "mask": list(map(int, m))} for i, t, m in zip(ids, types, mask)]


@pytest.mark.parametrize("variant", sorted(CASES["variants"]))
def test_golden_batches(variant):
    tok = HipTokenizer(_json_of(variant))
    v = CASES["variants"][variant]
    assert _rows(*tok.encode_batch(CASES["texts"])) == v["single_batch"]
    pairs = CASES["pairs"]
    assert _rows(*tok.encode_batch([p[0] for p in pairs], [p[1] for p in pairs])) == v["pair_batch"]
    for text, want in zip(CASES["texts"], v["single_each"]):
        assert _rows(*tok.encode_batch([text])) == [want], text


def test_capacity_error_reports_needed_length():
    tok = HipTokenizer(_json_of("template_24"))
    with pytest.raises(N.MrkError) as e:
        tok.encode_batch(["the quick brown fox jumps over the lazy dog"], capacity=4)
    assert e.value.status == N.ERR_INVALID_ARG


def test_rejects_other_pipelines():
    d = json.loads(_json_of("template_24"))
    d["model"]["type"] = "BPE"
    with pytest.raises(N.MrkError) as e:
        HipTokenizer(json.dumps(d))
    assert e.value.status == N.ERR_UNSUPPORTED
    with pytest.raises(N.MrkError) as e:
        HipTokenizer(b"{not json")
    assert e.value.status == N.ERR_PARSE


ALPHABETS = [
    "abcdefghijklmnopqrstuvwxyz", "ABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", " \t\n  ", ".,;:!?-()[]{}'\"/\\@#$%^&*+=<>|~`_",
    "àáâãäåæçèéêëìíîïñòóôõöùúûüýÿÀÉÎÕÜßŒœŠšŽžİıŁł", "αβγδεζηθικλμνξοπρςστυφχψωΑΒΓΔΣΩάέήίόύώϊϋΐΰ", "абвгдеёжзийклмнопрстуфхцчшщъыьэюяАБВГДЕЁЖЙ",
    "卧虎藏龙千尋神隠漢字仮名", "ひらがなカタカナｶﾀｶﾅ", "한국어영화가각", "אבגדהוזחט", "ابتثجحخدذ", "़ािीुूृेैोौ्कखगघ", "̸̧̨̛̣̀́̂̃̈",
    "–—‘’“”«»…•·¿¡§¶†‡‰′″‹›", "​‌‍⁠﻿­", "😀🎬⭐✓→∑∞≠", "ﬁﬂǅǈǋǲÅKΩ", "ẞİŉǰΐΰẖᾶ",
]


def _random_text(rng):
    n = int(rng.integers(0, 14))
    words = []
    for _ in range(n):
        alpha = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))] if rng.random() < 0.6 else ALPHABETS[0]
        ln = int(rng.integers(1, 9))
        w = "".join(alpha[int(rng.integers(0, len(alpha)))] for _ in range(ln))
        if rng.random() < 0.2:
            mark = ALPHABETS[14]
            w += mark[int(rng.integers(0, len(mark)))]
        words.append(w)
    sep = [" ", "  ", "-", ", ", "\t"][int(rng.integers(0, 5))]
    return sep.join(words)


@pytest.mark.parametrize("variant", ["template_24", "bertproc_17", "notrunc", "cased", "cased_strip"])
def test_live_against_huggingface_tokenizers(variant):
    tokenizers = pytest.importorskip("tokenizers")
    text = _json_of(variant).decode("utf-8")
    ref = tokenizers.Tokenizer.from_str(text)
    d = json.loads(text)
    if d.get("truncation") is None:
        ref.enable_truncation(max_length=512)
    if d.get("padding") is None:
        ref.enable_padding(pad_id=ref.token_to_id("[PAD]"), pad_token="[PAD]")
    tok = HipTokenizer(text)
    rng = np.random.default_rng(20250718)
    corpus = " ".join(CASES["texts"]).split()
    for rnd in range(40):
        texts = [_random_text(rng) if rng.random() < 0.7 else " ".join(rng.choice(corpus, size=int(rng.integers(1, 30)))) for _ in range(8)]
        want = ref.encode_batch(texts)
        got = _rows(*tok.encode_batch(texts))
        for t, w, g in zip(texts, want, got):
            assert g == {"ids": w.ids, "type_ids": w.type_ids, "mask": w.attention_mask}, repr(t)
        others = [" ".join(rng.choice(corpus, size=int(rng.integers(0, 25)))) for _ in range(8)]
        want = ref.encode_batch(list(zip(texts, others)))
        got = _rows(*tok.encode_batch(texts, others))
        for t, o, w, g in zip(texts, others, want, got):
            assert g == {"ids": w.ids, "type_ids": w.type_ids, "mask": w.attention_mask}, repr((t, o))
