"""Makes the text-encoder fixtures under tests/golden/ (run in the build container; needs torch, transformers,
tokenizers, safetensors -- all in the image; the GPU box only reads the committed outputs).

  tokenizer_tiny.json      a real HuggingFace tokenizer.json (BertWordPieceTokenizer trained on a synthetic corpus,
                           TemplateProcessing post-processor, truncation 24)
  tokenizer_cases.json     texts / pairs and the ids, type ids and masks the HuggingFace `tokenizers` library returns
                           for them (the library DJL's HuggingFaceTokenizer wraps -- OnnxSession.scala:42)
  encoder_tiny.onnx        torch.onnx export (the exporter behind the reference's `pytorch_model.onnx`) of a
                           random-init transformers.BertModel (2 layers, hidden 64, 2 heads)
  encoder_tiny.safetensors the same state_dict
  cross_tiny.onnx          BertForSequenceClassification(num_labels=1), same size
  encoder_tiny.npz         inputs and transformers' fp32 outputs: last_hidden_state, OnnxBiEncoder-style mean pool,
                           cross-encoder logits
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

CORPUS = [
    "The quick brown fox jumps over the lazy dog.", "Star Wars: Episode IV - A New Hope (1977)",
    "Amélie from Montmartre — café crème", "crouching tiger, hidden dragon 卧虎藏龙", "terminator 2 judgment day",
    "naïve façade coöperate", "İstanbul'da yağmur", "straße GROSS", "한국어 영화", "Ελληνικά ΣΟΦΟΣ", "résumé déjà vu",
    "the lord of the rings: the return of the king", "pulp fiction", "spirited away 千と千尋の神隠し", "$9.99 + 1/2 = ~10%",
]

TEXTS = [
    "Star Wars café", "the QUICK brown fox jumps over the lazy dog and the terminator again and again and again and again",
    "", "   ", "Amélie\tfrom\nMontmartre — café", "卧虎藏龙 crouching", "İstanbul STRASSE ΣΟΦΟΣ", "한국어", "áȩ́ ó̧",
    "hello [SEP] world [MASK]!", "unknownwordzzzzqqqq xylophone", "x" * 120 + " tail", "don't stop-believing (1981)...", "​zero‍widthnull�",
    "Ⅻ ﬁne Å K", "emoji 😀 ok", "¿qué? «quoted» 。、",
]
PAIRS = [
    ("star wars", "a new hope 1977 episode iv terminator judgment day dragon tiger lord of the rings return of the king"),
    ("x", "y"), ("the quick brown fox jumps over the lazy dog again and again and again and again and again", "fox"),
    ("", "pulp fiction"), ("spirited away", ""),
    ("the quick brown fox jumps over the lazy dog the quick brown fox", "the lord of the rings the return of the king pulp fiction spirited"),
    ("a b c d e f g h i j k l", "m n o p q r s t u v w"), ("a b c d e f g h i j k", "m n o p q r s t u v w x"),
]


def reinit(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "LayerNorm.weight" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "embeddings" in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p.shape[-1] ** 0.5))


def export(module, path, out_name):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # post-pass that needs the `onnx` package
    ids = torch.tensor([[2, 5, 7, 3, 0, 0]]); mask = torch.tensor([[1, 1, 1, 1, 0, 0]]); tt = torch.zeros_like(ids)
    ax = {0: "b", 1: "s"}
    torch.onnx.export(module, (ids, mask, tt), path, input_names=["input_ids", "attention_mask", "token_type_ids"],
                      output_names=[out_name], dynamic_axes={"input_ids": ax, "attention_mask": ax, "token_type_ids": ax},
                      dynamo=False, opset_version=14)


def main():
    from tokenizers import BertWordPieceTokenizer, Tokenizer, processors
    from transformers import BertConfig, BertForSequenceClassification, BertModel
    from safetensors.torch import save_file
    os.makedirs(OUT, exist_ok=True)

    t = BertWordPieceTokenizer(lowercase=True)
    t.train_from_iterator(CORPUS * 20, vocab_size=400, min_frequency=1, special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    tok = t._tokenizer
    cls, sep = tok.token_to_id("[CLS]"), tok.token_to_id("[SEP]")
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                       special_tokens=[("[CLS]", cls), ("[SEP]", sep)])
    tok.enable_truncation(max_length=24)
    tok.enable_padding()
    tok.save(os.path.join(OUT, "tokenizer_tiny.json"))

    def enc_cases(tk):
        s = tk.encode_batch(TEXTS)
        p = tk.encode_batch(PAIRS)
        one = [tk.encode_batch([x])[0] for x in TEXTS]
        f = lambda es: [{"ids": e.ids, "type_ids": e.type_ids, "mask": e.attention_mask} for e in es]
        return {"single_batch": f(s), "pair_batch": f(p), "single_each": f(one)}

    cases = {"texts": TEXTS, "pairs": [list(p) for p in PAIRS], "variants": {}}
    cases["variants"]["template_24"] = {"json": "tokenizer_tiny.json", **enc_cases(tok)}
    # variants of the same vocabulary: BertProcessing, no post-processor, no truncation entry, odd max_length, cased
    base = json.load(open(os.path.join(OUT, "tokenizer_tiny.json")))
    def variant(name, edit):
        d = json.loads(json.dumps(base)); edit(d)
        text = json.dumps(d, ensure_ascii=False)
        tk = Tokenizer.from_str(text)
        if d.get("truncation") is None:
            tk.enable_truncation(max_length=512)   # DJL truncation=true without a JSON entry: model max length
        if d.get("padding") is None:
            tk.enable_padding(pad_id=tk.token_to_id("[PAD]"), pad_token="[PAD]")
        cases["variants"][name] = {"json_text": text, **enc_cases(tk)}
    variant("bertproc_17", lambda d: (d.update(post_processor={"type": "BertProcessing", "sep": ["[SEP]", sep], "cls": ["[CLS]", cls]}),
                                      d["truncation"].update(max_length=17)))
    variant("nopost_9", lambda d: (d.update(post_processor=None), d["truncation"].update(max_length=9)))
    variant("notrunc", lambda d: d.update(truncation=None, padding=None))
    variant("cased", lambda d: d["normalizer"].update(lowercase=False, strip_accents=False))
    variant("cased_strip", lambda d: d["normalizer"].update(lowercase=False, strip_accents=True, handle_chinese_chars=False))
    json.dump(cases, open(os.path.join(OUT, "tokenizer_cases.json"), "w"), ensure_ascii=False, indent=0)

    V = tok.get_vocab_size()
    cfg = BertConfig(vocab_size=V, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, max_position_embeddings=48)
    bi = BertModel(cfg, add_pooling_layer=False).eval()
    reinit(bi, 1)

    class Hidden(torch.nn.Module):
        def __init__(s, m): super().__init__(); s.m = m
        def forward(s, input_ids, attention_mask, token_type_ids):
            return s.m(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids).last_hidden_state

    class Logits(torch.nn.Module):
        def __init__(s, m): super().__init__(); s.m = m
        def forward(s, input_ids, attention_mask, token_type_ids):
            return s.m(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids).logits

    export(Hidden(bi).eval(), os.path.join(OUT, "encoder_tiny.onnx"), "last_hidden_state")
    bi.eval()  # the exporter restores the wrapper's (training) mode on exit
    save_file({k: v.contiguous() for k, v in bi.state_dict().items()}, os.path.join(OUT, "encoder_tiny.safetensors"),
              metadata={"num_attention_heads": "2"})
    cfg1 = BertConfig(vocab_size=V, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, max_position_embeddings=48, num_labels=1)
    cross = BertForSequenceClassification(cfg1).eval()
    reinit(cross, 2)
    export(Logits(cross).eval(), os.path.join(OUT, "cross_tiny.onnx"), "logits")
    cross.eval()

    s = tok.encode_batch(TEXTS)
    ids = torch.tensor([e.ids for e in s]); tt = torch.tensor([e.type_ids for e in s]); mask = torch.tensor([e.attention_mask for e in s])
    p = tok.encode_batch(PAIRS)
    pids = torch.tensor([e.ids for e in p]); ptt = torch.tensor([e.type_ids for e in p]); pmask = torch.tensor([e.attention_mask for e in p])
    assert not bi.training and not cross.training
    with torch.no_grad():
        hidden = bi(input_ids=ids, attention_mask=mask, token_type_ids=tt).last_hidden_state.numpy()
        logits = cross(input_ids=pids, attention_mask=pmask, token_type_ids=ptt).logits[:, 0].numpy()
    from oracle import bert
    pooled = bert.avgpool(hidden, mask.numpy())
    np.savez_compressed(os.path.join(OUT, "encoder_tiny.npz"), ids=ids.numpy().astype(np.int32), type_ids=tt.numpy().astype(np.int32),
                        mask=mask.numpy().astype(np.int32), hidden=hidden, pooled=pooled, pair_ids=pids.numpy().astype(np.int32),
                        pair_type_ids=ptt.numpy().astype(np.int32), pair_mask=pmask.numpy().astype(np.int32), logits=logits)
    for f in sorted(os.listdir(OUT)):
        if "tiny" in f or "tokenizer" in f:
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
