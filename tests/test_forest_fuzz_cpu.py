"""CPU: the model readers and image packers (csrc/forest.cpp) under AddressSanitizer + UBSan against mutated model blobs
(tests/native/forest_fuzz.cpp): a corrupt `LambdaMARTModel` ends in an error status, never in memory corruption.
forest.cpp is host-only C++ - it is compiled here with g++ and the sanitizers, without the rest of the library."""
import os
import re
import struct
import subprocess

import numpy as np

from workloads import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "metarank_amd", "csrc")


def test_mutated_model_blobs_never_corrupt_memory(tmp_path):
    exe = str(tmp_path / "forest_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(REPO, "tests", "native", "forest_fuzz.cpp"), os.path.join(CSRC, "forest.cpp"), "-I" + CSRC, "-o", exe])
    q = np.linspace(-2.0, 2.0, 33)[None, :].repeat(6, 0)
    lgbm = synth.synthetic_lgbm_model(n_trees=6, n_features=6, num_leaves=16, max_depth=6, quantiles=q, cat_features=[2], cat_prob=0.2, missing="per_feature")
    seeds = {
        "lgbm:lgbm.txt": lgbm,
        "xgb:xgb.json": synth.synthetic_xgb_model(n_trees=5, n_features=6, depth=3, quantiles=q, cat_features=[2], cat_prob=0.2, fmt="json"),
        "xgb:xgb.ubj": synth.synthetic_xgb_model(n_trees=5, n_features=6, depth=3, quantiles=q, fmt="ubj"),
        "xgb:xgb.deep.json": synth.synthetic_xgb_model(n_trees=3, n_features=6, depth=6, quantiles=q, fmt="json", complete=False),
        "xgb:xgb.legacy": synth.synthetic_xgb_model(n_trees=5, n_features=6, depth=3, quantiles=q, fmt="legacy"),
        "container:lgbm.container": synth.write_container([f"f{i}" for i in range(6)], 0, lgbm, version=3, warmup=0),
        "container:xgb.container": synth.write_container([f"f{i}" for i in range(6)], 1, synth.synthetic_xgb_model(
            n_trees=3, n_features=6, depth=3, quantiles=q, fmt="json"), version=2),
    }
    # structured mutants as seeds too: another key order and unknown keys at every level (tests/test_model_loaders_cpu.py), then the
    # byte-level mutations on top of them
    import json
    import random
    from test_model_loaders_cpu import _shuffled, shuffled_lgbm
    rng = random.Random(3)
    seeds["xgb:xgb.shuffled.json"] = json.dumps(_shuffled(json.loads(seeds["xgb:xgb.json"]), rng, ["zz_new", "stats"])).encode()
    seeds["lgbm:lgbm.shuffled.txt"] = shuffled_lgbm(lgbm.decode() if isinstance(lgbm, (bytes, bytearray)) else lgbm, rng).encode()
    args = []
    for name, blob in seeds.items():
        kind, fname = name.split(":")
        path = tmp_path / fname
        path.write_bytes(blob if isinstance(blob, (bytes, bytearray)) else blob.encode())
        args.append(f"{kind}:{path}")
    # the defects the first mutation runs found, as crafted blobs that must be refused (each was memory-unsafe or undefined
    # behaviour before its fix)
    xgb = seeds["xgb:xgb.json"].decode()
    lgbm_text = lgbm.decode() if isinstance(lgbm, (bytes, bytearray)) else lgbm
    crafted = {
        "lgbm:num_cat_int_max": lgbm_text.replace("num_cat=3", "num_cat=2147483647", 1),                       # signed overflow in num_cat + 1
        "lgbm:max_feature_idx": lgbm_text.replace("max_feature_idx=5", "max_feature_idx=2147483647"),          # ... in max_feature_idx + 1
        "lgbm:cat_boundaries": lgbm_text.replace("cat_boundaries=0 1 2 3", "cat_boundaries=0 1 2 4294967295", 1),
        "xgb:category_2_40": xgb.replace('"categories":[3,', '"categories":[1099511627776,', 1),           # OOB write: 64-bit index cut to 32 bits
        "xgb:negative_segment": xgb.replace('"categories_segments":[0,', '"categories_segments":[-1,', 1),  # OOB read: b + sz wrapped
        "xgb:huge_size": xgb.replace('"categories_sizes":[7,', '"categories_sizes":[9223372036854775807,', 1),
        "xgb:num_feature": xgb.replace('"num_feature":"6"', '"num_feature":"4294967296"'),
        # UBJSON: {"learner": [ <count = 2^31 - 1, no elements> : the count sized an allocation before the input was looked at
        "xgb:ubj_count": b"{i\x07learner[#l" + struct.pack(">i", 2**31 - 1) + b"}",
        "xgb:ubj_typed_count": b"{i\x07learner[$d#L" + struct.pack(">q", 2**40) + b"}",
    }
    for name, blob in crafted.items():
        kind, fname = name.split(":")
        data = blob if isinstance(blob, (bytes, bytearray)) else blob.encode()
        assert data not in [v if isinstance(v, (bytes, bytearray)) else v.encode() for v in seeds.values()], name  # the pattern was found
        path = tmp_path / ("crafted_" + fname)
        path.write_bytes(data)
        args.append(f"reject-{kind}:{path}")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, "2500"] + args, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:] + out.stderr[-6000:])
    assert "survived 22500 mutants" in out.stdout, out.stdout
    assert out.stdout.count(": rejected") == len(crafted), out.stdout
