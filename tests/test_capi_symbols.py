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
    assert L.mrk_abi_version() == 9
    assert L.mrk_model_weights(None, 1, None, 0) == _native.ERR_INVALID_ARG   # ABI 9
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


def test_abi_layout_matches_the_header_and_the_ctypes_structures(tmp_path):
    """mrk_abi_layout(): what a JNA @FieldOrder binding checks at start-up.  Three independent views of the same
    structs must agree: the library's own numbers, the ctypes Structures of the Python harness, and a C99 program compiled from
    include/mrk.h by gcc."""
    import ctypes as C
    import subprocess

    L = _native.lib()
    n = L.mrk_abi_layout(None, 0)
    v = (C.c_int32 * n)()
    assert L.mrk_abi_layout(v, n) == n and n == 33 and v[0] == L.mrk_abi_version()
    short = (C.c_int32 * 3)(-1, -1, -1)
    assert L.mrk_abi_layout(short, 2) == n and short[2] == -1   # never writes past cap
    def view(st, fields):
        return [C.sizeof(st)] + [getattr(st, f).offset for f in fields]
    fields = {"mrk_field": ["name", "type", "n", "num", "str", "strs", "nums"],
              "mrk_request": ["id", "timestamp_ms", "user", "session", "fields", "n_fields", "n_items", "item_ids", "item_field_offsets", "item_fields"],
              "mrk_model_info": ["backend", "n_trees", "max_depth", "n_features", "is_f64", "n_categorical", "n_nodes", "n_leaves", "device_bytes",
                                 "base_score", "bitvector", "tile_columns"]}
    got = list(v)
    want = [L.mrk_abi_version()] + view(_native.mrk_field, fields["mrk_field"]) + view(_native.mrk_request, fields["mrk_request"]) + \
        view(_native.mrk_model_info, fields["mrk_model_info"])
    assert got == want
    # every field of the three structs is covered (a field added to the header without a layout entry fails here)
    for name, fs in fields.items():
        assert [f for f, _ in getattr(_native, name)._fields_] == fs
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mrk.h"', 'int main(void) {', '  printf("%d", MRK_ABI_VERSION);']
    for name, fs in fields.items():
        lines.append(f'  printf(" %zu", sizeof({name}));')
        lines += [f'  printf(" %zu", offsetof({name}, {f}));' for f in fs]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    assert [int(x) for x in subprocess.check_output([str(exe)]).split()] == got


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
