"""CPU: the model readers and image packers (csrc/forest.cpp) under AddressSanitizer + UBSan against mutated model blobs
(tests/native/forest_fuzz.cpp): a corrupt `LambdaMARTModel` ends in an error status, never in memory corruption.
forest.cpp is host-only C++ - it is compiled here with g++ and the sanitizers, without the rest of the library."""
import os
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
    args = []
    for name, blob in seeds.items():
        kind, fname = name.split(":")
        path = tmp_path / fname
        path.write_bytes(blob if isinstance(blob, (bytes, bytearray)) else blob.encode())
        args.append(f"{kind}:{path}")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, "2500"] + args, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:] + out.stderr[-6000:])
    assert "survived 17500 mutants" in out.stdout, out.stdout
