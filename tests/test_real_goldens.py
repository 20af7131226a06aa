"""Golden vectors written by the REAL LightGBM / XGBoost (tools/make_real_goldens.py, tests/golden/README.md): the
oracle and the HIP scorers must reproduce the libraries' own predictions bit for bit.  The fixtures cannot be generated
in the build image (neither library is installed, no network), so these tests skip until someone drops them into
tests/golden/real/."""
import glob
import os

import numpy as np
import pytest

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real")
CASES = sorted(f[:-4] for f in glob.glob(os.path.join(REAL, "*.npz")) if not f.endswith("minilm.npz"))
needs_fixtures = pytest.mark.skipif(not CASES, reason="no real-library goldens in tests/golden/real (see tests/golden/README.md)")


def _load(case):
    z = np.load(case + ".npz", allow_pickle=False)
    blob = open(case + ".model", "rb").read()
    return blob, z["X"], z["pred"], int(z["backend"])


@needs_fixtures
@pytest.mark.parametrize("case", CASES or ["none"])
def test_oracle_reproduces_the_library(case):
    from oracle.forest import OracleForest

    blob, X, pred, backend = _load(case)
    f = OracleForest.from_lightgbm_text(blob) if backend == 0 else OracleForest.from_xgboost(blob)
    got = f.predict(X)
    want = pred.astype(np.float64)   # XGBoost: the f32 margin widened, as ltrlib returns it
    assert np.array_equal(got, want, equal_nan=True), float(np.nanmax(np.abs(got - want)))


@needs_fixtures
@pytest.mark.parametrize("case", CASES or ["none"])
def test_weights_reproduce_the_library_importances(case):
    """mrk_model_inspect_weights (== Booster.weights(), include/mrk.h) against the library's own feature importances, bit for bit
    (fixtures written before round 6 carry none: skipped per case)."""
    from metarank_amd.booster import inspect_weights

    z = np.load(case + ".npz", allow_pickle=False)
    if "importance_gain" not in z.files:
        pytest.skip("fixture without importances: re-run tools/make_real_goldens.py")
    blob, backend = open(case + ".model", "rb").read(), int(z["backend"])
    n = len(z["importance_gain"])
    for kind, key in ((0, "importance_split"), (1, "importance_gain"), (2, "importance_total_gain")):
        if key in z.files:
            assert np.array_equal(inspect_weights(blob, backend, n, kind), z[key]), key


@needs_fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES or ["none"])
def test_hip_reproduces_the_library(case):
    import metarank_amd as M

    blob, X, pred, backend = _load(case)
    b = M.HipBooster(blob, backend)
    assert np.array_equal(b.predict(X), pred.astype(np.float64), equal_nan=True)
    b.close()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REAL, "minilm.npz")) or not os.environ.get("MRK_MINILM_DIR"),
                    reason="needs tests/golden/real/minilm.npz and MRK_MINILM_DIR=<dir with pytorch_model.onnx + tokenizer.json>")
def test_minilm_matches_the_reference_numbers():
    """OnnxBiencoderTest.scala:13-25: cosine 0.539 / 0.738 +- 1e-3 with the reference's own all-MiniLM-L6-v2 export"""
    from metarank_amd.encoder import HipEncoder

    d = os.environ["MRK_MINILM_DIR"]
    z = np.load(os.path.join(REAL, "minilm.npz"))
    enc = HipEncoder(open(os.path.join(d, "pytorch_model.onnx"), "rb").read(), open(os.path.join(d, "tokenizer.json"), "rb").read())
    texts = [str(t) for t in z["texts"]]
    e = np.concatenate([enc.embed(texts[:1]), enc.embed(texts[1:])])

    def cos(a, b):
        b = b.astype(np.float64)
        return float((a.astype(np.float64) * b).sum() / (np.sqrt((a * a).astype(np.float64).sum()) * np.sqrt((b * b).sum())))
    assert abs(cos(e[1], e[2]) - 0.539) < 1e-3 and abs(cos(e[1], e[3]) - 0.738) < 1e-3
    assert np.abs(e - z["embeddings"]).max() < 3e-3   # fp16 operands vs onnxruntime's f32 graph
    enc.close()
