"""CPU: libmrk_hip.so loads and exports every symbol include/mrk.h declares (no compute calls)."""
import pytest
import ctypes as C
import os
import re

from metarank_amd import _native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mrk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    _native.build()
    L = C.CDLL(_native.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mrk.h but not exported"
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_native.SIGNATURES) == set(syms)


def test_abi_version_and_error_paths_without_gpu():
    L = _native.lib()
    assert L.mrk_abi_version() == 8
    # null arguments are rejected before any device work
    assert L.mrk_model_predict_f64(None, None, 1, 1, None) == _native.ERR_INVALID_ARG
    assert b"null model" in L.mrk_last_error()
    L.mrk_model_free(None)
    L.mrk_shutdown(None)
    # round 3's entry points: argument checks come before any device work
    import ctypes as C
    out = C.c_void_p()
    assert L.mrk_serve_start(None, None, b"m", 2, C.byref(out)) == _native.ERR_INVALID_ARG and not out.value
    assert L.mrk_serve_rank(None, None, None, None) == _native.ERR_INVALID_ARG
    assert L.mrk_serve_stats(None, None, 0) == _native.ERR_INVALID_ARG
    L.mrk_serve_stop(None)
    assert L.mrk_encoder_load_ex(None, None, 0, None, 0, 0, 1, C.byref(out)) == _native.ERR_INVALID_ARG
    assert L.mrk_config_warmup(None, b"m") == _native.ERR_INVALID_ARG
    assert L.mrk_shard_chunk(1000, 8) == 128 and L.mrk_shard_chunk(1025, 8) == 256
    # ABI 7: what a measurement records to say which code it ran
    bid = L.mrk_build_id()
    assert re.fullmatch(rb"[0-9a-f]{16}", bid), bid
    need = C.c_size_t(0)
    assert L.mrk_config_kernel_keys(None, b"m", None, 0, C.byref(need)) == _native.ERR_INVALID_ARG
    assert L.mrk_config_precompile_for_model(None, 0, b"m", 0, None, 0, 1, b"/tmp", None) == _native.ERR_INVALID_ARG
    assert L.mrk_config_specialize_for_model(b"{}", 2, b"m", 0, None, 0, 0, None, 0, C.byref(need)) == _native.ERR_INVALID_ARG   # no model bytes


def test_build_id_follows_the_sources(tmp_path):
    """mrk_build_id() is the digest _native.write_build_id() takes over csrc/ and include/mrk.h when the library is built: the
    library on disk was built from the sources on disk (bench.py quotes profiler counters only for the build they were taken on)."""
    import hashlib

    csrc = os.path.join(REPO, "metarank_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".hpp")):
            h.update(f.encode() + b"\0" + open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(REPO, "include", "mrk.h"), "rb").read())
    _native.build()
    assert _native.lib().mrk_build_id().decode() == h.hexdigest()[:16]


def test_no_product_file_references_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "metarank_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|liboracle", src, flags=re.M):
                    bad.append(f)
    assert bad == []


def test_header_is_plain_c99():
    """include/mrk.h is what a JNA / cgo / ctypes binding is written against: it must compile as C (no C++ isms, no torch
    types), alone, with -std=c99 -pedantic."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(REPO, "include", "mrk.h")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
