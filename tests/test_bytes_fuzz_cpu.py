"""CPU: the other byte formats the library reads from outside - encoder checkpoints (safetensors, ONNX protobuf), the
binary RankingEventFormat, the binary FeatureValue stream, tokenizer.json, the feature / model configuration - under
AddressSanitizer + UBSan against mutated blobs (tests/native/bytes_fuzz.cpp).  The readers (csrc/weights.cpp, codec.cpp,
tokenizer.cpp, features.cpp) are compiled into the test binary with the sanitizers; what they call comes from
libmrk_hip.so.  No device."""
import json
import os
import shutil
import struct
import subprocess

from oracle import codec
from workloads import ranklens, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "metarank_amd", "csrc")


def test_mutated_checkpoints_requests_and_feature_values_never_corrupt_memory(tmp_path):
    from metarank_amd import _native

    _native.build()
    lib_dir = os.path.dirname(_native.LIB_PATH)
    exe = str(tmp_path / "bytes_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(REPO, "tests", "native", "bytes_fuzz.cpp"),
                           os.path.join(CSRC, "weights.cpp"), os.path.join(CSRC, "codec.cpp"), os.path.join(CSRC, "tokenizer.cpp"),
                           os.path.join(CSRC, "features.cpp"), "-I" + CSRC, "-I" + os.path.join(REPO, "include"),
                           "-L" + lib_dir, "-lmrk_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    w = synth.synthetic_bert(layers=1, hidden=64, heads=2, inter=128, vocab=50, max_pos=16, classifier=True)
    (tmp_path / "enc.safetensors").write_bytes(synth.bert_safetensors(w, 2))
    shutil.copy(os.path.join(REPO, "tests", "golden", "encoder_tiny.onnx"), tmp_path / "enc.onnx")
    reqs = ranklens.generate_requests(2, 5, 50, 10, seed=3)
    reqs[0]["fields"] = [{"name": "query", "value": "red socks"}, {"name": "n", "value": 3.5}, {"name": "flags", "value": ["a", "b"]},
                         {"name": "ok", "value": True}, {"name": "v", "value": [1.0, 2.0]}]
    reqs[0]["items"][0]["fields"] = [{"name": "popularity", "value": 7.5}, {"name": "genres", "value": ["x", "y"]}]
    (tmp_path / "req.bin").write_bytes(b"".join(codec.ranking_event(ev) for ev in reqs))
    values = [("string", "item=i1/genre", "a"), ("string_list", "item=i1/tags", ["a", "b"]), ("double_list", "item=i1/vec", [1.5, -2.0, 3.0]),
              ("double", "item=i1/pop", 2.5), ("counter", "item=i1/clicks", 5), ("periodic", "item=i1/ctr_click", [1, 2, 3]),
              ("bounded_list", "session=s1/profile", ["i1", "i2"]), ("numstats", "item=i1/x", (0.0, 1.0, {50: 0.5})),
              ("map", "item=i1/m", {"k": ("double", 1.0)}), ("freq", "item=i1/f", {"k": 0.5})]
    (tmp_path / "fv.bin").write_bytes(b"".join(codec.feature_value(kind, key, v, compat=(i % 3 == 0)) for i, (kind, key, v) in enumerate(values)))
    (tmp_path / "tok.json").write_text(synth.wordpiece_tokenizer_json(vocab_size=200, max_length=32))
    (tmp_path / "config.json").write_text(json.dumps(ranklens.ranklens_config()))
    # crafted blobs for the defects the first mutation runs found: must be refused
    tok = (tmp_path / "tok.json").read_text()
    cfg = (tmp_path / "config.json").read_text()
    crafted = {
        # a ranking event that declares 2^31 - 1 fields and ends: the count sized an allocation (51 GB) before the input was read
        "request:field_count": codec.utf("r") + struct.pack(">q", 0) + b"\x00\x00" + struct.pack(">i", 2**31 - 1),
        "tok:max_length": tok.replace('"max_length": 32', '"max_length": -2147483648'),     # signed overflow in the truncation
        "config:bucket": cfg.replace('"bucket": "24h"', '"bucket": "999999999999999999d"', 1),  # ... in the duration
    }
    assert crafted["tok:max_length"] != tok and crafted["config:bucket"] != cfg
    extra = []
    for name, blob in crafted.items():
        kind, fname = name.split(":")
        path = tmp_path / ("crafted_" + fname)
        path.write_bytes(blob if isinstance(blob, (bytes, bytearray)) else blob.encode())
        extra.append(f"reject-{kind}:{path}")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, "1500", f"ckpt:{tmp_path / 'enc.safetensors'}", f"ckpt:{tmp_path / 'enc.onnx'}", f"request:{tmp_path / 'req.bin'}",
                          f"fv:{tmp_path / 'fv.bin'}", f"tok:{tmp_path / 'tok.json'}", f"config:{tmp_path / 'config.json'}"] + extra, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:] + out.stderr[-6000:])
    assert "survived 9000 mutants" in out.stdout, out.stdout
    assert out.stdout.count(": rejected") == len(crafted), out.stdout
