"""CPU: the run-time specialised assembly kernel (metarank_amd/csrc/jit.cpp) can be generated and compiled for gfx950
without a device: mrk_config_specialize returns the translation unit hiprtc gets (the shared device code + the model's
program as constants) and its code object.  Whether that kernel computes the right thing is a GPU test
(tests/test_rank_parity.py::test_assembly_paths_agree runs it against the generic kernel and the oracle)."""
import ctypes as C
import json

import pytest

from metarank_amd import _native
from workloads import ranklens


def specialize(cfg, what, f64=1, model=b"xgboost"):
    lib = _native.lib()
    js = json.dumps(cfg).encode()
    need = C.c_size_t(0)
    rc = lib.mrk_config_specialize(js, len(js), model, f64, what, None, 0, C.byref(need))
    if rc == _native.ERR_INVALID_ARG and need.value:
        buf = (C.c_uint8 * need.value)()
        rc = lib.mrk_config_specialize(js, len(js), model, f64, what, buf, need.value, C.byref(need))
        return rc, bytes(buf[:need.value])
    return rc, b""


def test_source_carries_the_program_as_constants():
    cfg = ranklens.ranklens_config()
    rc, src = specialize(cfg, 0)
    assert rc == 0
    text = src.decode()
    # 18 features of the stock Ranklens model -> 18 ops, 24 matrix columns, 4 interacted_with fields + 5 diversity reductions
    assert "n_ops = 18, n_prep = 9, dim = 24" in text
    assert "struct JitOps" in text and "rank_fused_cells_body<true, false, mrk::QsDyn>" in text and "mrk_jit_rank_cells_split" in text and "#include" not in text
    rc, src32 = specialize(cfg, 0, f64=0)
    assert rc == 0 and b"rank_fused_cells_body<false, false, mrk::QsDyn>" in src32
    # the normalised rate's weight travels as an exact hexadecimal literal (10.0)
    assert "0x1.4p+3" in text
    # what the library compiles by itself is ONE kernel per translation unit (the kernel a batch shape needs, 10 - 20 s of
    # compiler time each instead of 50 s for all four): what = form | (1 + kernel) << 8
    names = ["mrk_jit_rank_cells(", "mrk_jit_rank_cells_split(", "mrk_jit_rank_matrix(", "mrk_jit_assemble_cells(", "mrk_jit_rank_one(",
             "mrk_jit_rank_serve(", "mrk_jit_rank_fused_score("]
    flat = text.replace("\n", "")
    assert all(n in flat for n in names)
    for k, name in enumerate(names):
        rc, one = specialize(cfg, 0 | ((k + 1) << 8))
        one = one.decode().replace("\n", "")
        assert rc == 0 and name in one and not any(o in one for o in names if o != name), name
    # the resident-table form of the item-parallel kernel (jit.hpp JIT_ITEMS_RT) exists only with a forest signature
    rc, rt = specialize(cfg, 0 | (9 << 8))
    assert rc == 0 and b"mrk_jit_assemble_cells_rt(" not in rt.replace(b"\n", b"")
    assert specialize(cfg, 0 | (10 << 8))[0] == _native.ERR_INVALID_ARG


def specialize_for_model(cfg, backend, blob, what, model=b"xgboost"):
    lib = _native.lib()
    js = json.dumps(cfg).encode()
    need = C.c_size_t(0)
    rc = lib.mrk_config_specialize_for_model(js, len(js), model, backend, blob, len(blob), what, None, 0, C.byref(need))
    if rc == _native.ERR_INVALID_ARG and need.value:
        buf = (C.c_uint8 * need.value)()
        rc = lib.mrk_config_specialize_for_model(js, len(js), model, backend, blob, len(blob), what, buf, need.value, C.byref(need))
        return rc, bytes(buf[:need.value])
    return rc, b""


def test_source_carries_the_forests_view_signature():
    """The kernels that write the scorer's tile are keyed by the forest too: per column its views and the chunks of its
    threshold table, as constants (forest.hpp QsSignature) - not the thresholds."""
    import re

    from workloads import synth

    cfg = ranklens.ranklens_config()
    blob = synth.synthetic_lgbm_model(n_trees=40, n_features=24, missing="per_feature", cat_features=[7], cat_prob=0.05)
    rc, src = specialize_for_model(cfg, 0, blob, 0 | (1 << 8))
    assert rc == 0, _native.lib().mrk_last_error()
    text = src.decode()
    assert "struct JitQs" in text and "rank_fused_cells_body<true, false, mrk::JitQs>" in text and "n_feats = 24" in text
    rows = re.search(r"struct JitSigRows \{.*?= \{(.*?)\};", text, re.S).group(1)
    assert rows.count("{") == 24
    # the resident-table item-parallel kernel: every threshold table in LDS, their total size a constant of the signature
    rc, rsrc = specialize_for_model(cfg, 0, blob, 0 | (9 << 8))
    assert rc == 0 and b"mrk_jit_assemble_cells_rt(" in rsrc.replace(b"\n", b"") and b"assemble_cells_rt_body<true, mrk::JitQs>" in rsrc
    total = int(re.search(r"thr_total = (\d+)u", rsrc.decode()).group(1))
    compact = int(re.search(r"rt_total = (\d+)u", rsrc.decode()).group(1))   # the same tables padded to powers of two instead of 128-entry chunks
    assert total % 128 == 0 and compact % 2 == 0 and 0 < compact <= total and compact * 8 <= 64 * 1024
    # ... and every column's place in the compact layout: exact lengths, back to back in column order (columns the forest never
    # splits on take no room), at most 256 entries resident
    sig_rows = [tuple(int(x.rstrip("u")) for x in r.split(",")) for r in re.findall(r"\{([0-9u,]+)\}", re.search(r"struct JitSigRows \{.*?= \{(.*?)\};", rsrc.decode(), re.S).group(1))]
    assert len(sig_rows) == 24 and all(len(r) == 7 for r in sig_rows)
    at = 0
    for thr_off, chunks, vb, ve, kinds, rt_off, rt_len in sig_rows:
        assert thr_off % 128 == 0 and rt_len <= min(256, chunks * 128)
        if vb != ve:
            assert rt_off >= at
            at = rt_off + rt_len
    assert at <= compact <= at + 1
    # the f64-matrix kernel bins nothing: no signature in its translation unit
    rc, msrc = specialize_for_model(cfg, 0, blob, 0 | (3 << 8))
    assert rc == 0 and b"JitQs" not in msrc
    # another forest over the same columns with the same missing-value kinds and table sizes: the same text, hence the same kernel
    blob2 = synth.synthetic_lgbm_model(n_trees=40, n_features=24, missing="per_feature", cat_features=[7], cat_prob=0.05, seed=11)
    rc, src2 = specialize_for_model(cfg, 0, blob2, 0 | (1 << 8))
    assert rc == 0
    same_sig = re.search(r"struct JitSigRows \{.*?= \{(.*?)\};", src2.decode(), re.S).group(1) == rows
    assert (src2 == src) == same_sig
    # XGBoost: f32 precision follows from the backend
    xblob = synth.synthetic_xgb_model(n_trees=10, n_features=24, depth=3)
    rc, xsrc = specialize_for_model(cfg, 1, xblob, 0 | (1 << 8))
    assert rc == 0 and b"rank_fused_cells_body<false, false, mrk::JitQs>" in xsrc
    assert specialize_for_model(cfg, 0, b"not a model", 0)[0] == _native.ERR_PARSE
    assert specialize_for_model(cfg, 7, blob, 0)[0] == _native.ERR_INVALID_ARG


@pytest.mark.parametrize("backend", [0, 1])
def test_hiprtc_compiles_the_signature_keyed_kernels(backend):
    from workloads import synth

    cfg = ranklens.ranklens_config()
    blob = (synth.synthetic_lgbm_model(n_trees=40, n_features=24, missing="per_feature", cat_features=[7], cat_prob=0.05) if backend == 0
            else synth.synthetic_xgb_model(n_trees=10, n_features=24, depth=3))
    for kernel in (1, 5, 9):   # the kernel of full batches, mrk_rank's one-launch kernel, the resident-table item-parallel kernel
        rc, code = specialize_for_model(cfg, backend, blob, 1 | (kernel << 8))
        assert rc == 0, _native.lib().mrk_last_error()
        assert code[:4] == b"\x7fELF" and b"gfx950" in code


def test_unknown_model_and_bad_arguments():
    lib = _native.lib()
    rc, _ = specialize(ranklens.ranklens_config(), 0, model=b"nope")
    assert rc == _native.ERR_NOT_FOUND
    assert lib.mrk_config_specialize(None, 0, b"x", 1, 0, None, 0, None) == _native.ERR_INVALID_ARG
    need = C.c_size_t(0)
    assert lib.mrk_config_specialize(b"{", 1, b"x", 1, 0, None, 0, C.byref(need)) == _native.ERR_PARSE


@pytest.mark.parametrize("which,f64", [("c2", 1), ("c3", 0)])
def test_hiprtc_compiles_the_specialised_kernel_for_gfx950(which, f64):
    cfg = ranklens.c3_config() if which == "c3" else ranklens.ranklens_config()
    kernel = 1 if which == "c2" else 2  # c2: the kernel of full batches; c3: the sliced form its batches run
    rc, code = specialize(cfg, 1 | (kernel << 8), f64=f64)
    assert rc == 0, _native.lib().mrk_last_error()
    assert code[:4] == b"\x7fELF" and b"gfx950" in code
    assert (b"mrk_jit_rank_cells_split" in code) == (kernel == 2) and b"mrk_jit_rank_cells" in code and b"mrk_jit_assemble_cells" not in code


def test_compiles_when_torch_brought_its_own_rocm_libraries(tmp_path):
    """bench.py --gpus N > 1 imports torch (the RCCL binding) before the library; the PyTorch wheel bundles its own
    libhiprtc / libamd_comgr / libamdhip64 under the same sonames, so those are what libmrk_hip.so then runs on.  The
    specialised kernel must compile there too (own process: the import order is the point)."""
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import torch, ctypes as C, json, os, sys
sys.path.insert(0, %r)
from metarank_amd import _native
from workloads import ranklens
lib = _native.lib()
js = json.dumps(ranklens.ranklens_config()).encode()
need = C.c_size_t(0)
lib.mrk_config_specialize(js, len(js), b"xgboost", 1, 1 | (3 << 8), None, 0, C.byref(need))
buf = (C.c_uint8 * need.value)()
rc = lib.mrk_config_specialize(js, len(js), b"xgboost", 1, 1 | (3 << 8), buf, need.value, C.byref(need))
maps = open("/proc/%%d/maps" %% os.getpid()).read()
print("RC", rc, bytes(buf[:4]) == b"\x7fELF", "torch/lib/libhiprtc" in maps)
""" % repo
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MRK_JIT_CACHE_DIR=str(tmp_path)))
    assert "RC 0 True" in out.stdout, out.stdout + out.stderr


def test_precompile_writes_the_code_objects_a_deployment_ships(tmp_path):
    """mrk_config_precompile (host-only): one gfx950 code object per kernel of the mask into a directory - what
    __graft_entry__.build() puts next to the library for the stock Ranklens program (metarank_amd/jit_cache) so that a process
    serving that model never compiles.  A small program here (one feature: seconds per kernel); a second call finds the files."""
    import os

    from backends import single_feature_config

    lib = _native.lib()
    cfg = single_feature_config({"name": "popularity", "type": "number", "scope": "item", "source": "item.popularity"})
    js = json.dumps({"features": cfg["features"], "models": cfg["models"]}).encode()
    model = list(cfg["models"])[0].encode()
    n = C.c_int(-1)
    mask = (1 << 0) | (1 << 4)   # the full-batch kernel and mrk_rank's one-launch kernel
    assert lib.mrk_config_precompile(js, len(js), model, 1, mask, str(tmp_path).encode(), C.byref(n)) == 0, lib.mrk_last_error()
    files = sorted(os.listdir(tmp_path))
    assert n.value == 2 and len(files) == 2 and all(f.endswith("-gfx950.co") for f in files)
    for f in files:
        blob = open(os.path.join(tmp_path, f), "rb").read()
        assert blob[:4] == b"\x7fELF" and (b"mrk_jit_rank_cells" in blob or b"mrk_jit_rank_one" in blob)
    assert lib.mrk_config_precompile(js, len(js), model, 1, mask, str(tmp_path).encode(), C.byref(n)) == 0 and n.value == 0
    assert lib.mrk_config_precompile(js, len(js), b"nope", 1, mask, str(tmp_path).encode(), C.byref(n)) == _native.ERR_NOT_FOUND
    assert lib.mrk_config_precompile(None, 0, model, 1, mask, str(tmp_path).encode(), None) == _native.ERR_INVALID_ARG
