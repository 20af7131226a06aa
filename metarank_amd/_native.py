"""ctypes binding of libmrk_hip.so (include/mrk.h).  Plumbing only: the product is the library."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("MRK_LIB") or os.path.join(HERE, "libmrk_hip.so")  # MRK_LIB: another build of the library (same-box A/B runs)
SOURCES = ["forest.cpp", "store.cpp", "features.cpp", "codec.cpp", "tokenizer.cpp", "weights.cpp", "capi.cpp", "capi_rank.cpp", "capi_encoder.cpp", "comm.cpp", "jit.cpp", "score.hip", "score_qs.hip", "rank.hip", "resolve.hip", "writes.hip", "encoder.hip"]
HEADERS = ["json.hpp", "forest.hpp", "runtime.hpp"]

MRK_OK = 0
ERR_INVALID_ARG, ERR_PARSE, ERR_DEVICE, ERR_DIM_MISMATCH = -1, -2, -3, -4
ERR_ARITHMETIC, ERR_UNSUPPORTED, ERR_NOT_FOUND, ERR_FEATURE_MISMATCH = -5, -6, -7, -8


class MrkError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[{status}] {message}")
        self.status = status
        self.message = message


def sources():
    extra = sorted(f for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")) and f not in SOURCES)
    return [os.path.join(CSRC, f) for f in SOURCES + extra]


# device headers that make up the translation unit of the run-time specialised assembly kernel (csrc/jit.cpp), in
# include order
JIT_HEADERS = ["device_types.hpp", "rank.hpp", "qs_device.hpp", "sort_device.hpp", "table_device.hpp", "wave_device.hpp", "rank_device.hpp"]


def embed_jit_sources() -> str:
    """csrc/jit_embed.inc: the JIT_HEADERS without their #include / #pragma once lines as C++ raw string literals (the
    text hiprtc compiles, followed at run time by the model's program as constants).  Generated, not committed."""
    parts = []
    for h in JIT_HEADERS:
        lines = [ln for ln in open(os.path.join(CSRC, h)).read().split("\n")
                 if not ln.startswith("#include") and not ln.startswith("#pragma once")]
        parts.append(f"// ---- {h}\n" + "\n".join(lines))
    text = "\n".join(parts)
    assert ')MRKJIT"' not in text
    # one literal per header: compilers cap the length of a single string literal, adjacent literals concatenate
    out = "\n".join('R"MRKJIT(' + p + '\n)MRKJIT"' for p in parts) + "\n"
    path = os.path.join(CSRC, "jit_embed.inc")
    if not os.path.exists(path) or open(path).read() != out:
        with open(path, "w") as f:
            f.write(out)
    return path


def write_build_id() -> str:
    """csrc/build_id.inc: a string literal with the digest of every source of the library (csrc/ without the generated
    files, include/mrk.h) - what mrk_build_id() returns.  Generated, not committed; only capi.cpp includes it."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".cpp", ".hip", ".hpp")):
            h.update(f.encode() + b"\0" + open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(REPO, "include", "mrk.h"), "rb").read())
    out = '"' + h.hexdigest()[:16] + '"\n'
    path = os.path.join(CSRC, "build_id.inc")
    if not os.path.exists(path) or open(path).read() != out:
        with open(path, "w") as f:
            f.write(out)
    return path


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every source into metarank_amd/libmrk_hip.so (in-tree).  One object per source under
    metarank_amd/build/ (compiled in parallel, rebuilt when the source or any header is newer), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    embed_jit_sources()
    build_id = write_build_id()
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")) and f != "build_id.inc"] + [os.path.join(REPO, "include", "mrk.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    defines = [f"-D{d}" for d in os.environ.get("MRK_DEFINES", "").split()]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-ffp-contract=off"] + defines  # the JVM never fuses a*b+c; parity with the reference is bit-exact
    objdir = os.environ.get("MRK_BUILD_DIR") or os.path.join(HERE, "build")  # MRK_BUILD_DIR + MRK_LIB: a second build (other defines) beside the default one
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        newest = max(os.path.getmtime(src), hdr_time, os.path.getmtime(build_id) if src.endswith("capi.cpp") else 0)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            subprocess.check_call([hipcc] + flags + ["-c", src, "-o", obj])
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max(1, min(len(srcs), os.cpu_count() or 1))) as ex:
        done = list(ex.map(compile_one, srcs))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    objs = [o for o, _ in done]
    if any(c for _, c in done) or not os.path.exists(LIB_PATH):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs +
                              ["-L/opt/rocm/lib", "-lhiprtc", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])  # hiprtc: csrc/jit.cpp; rccl: csrc/comm.cpp
    return LIB_PATH


class mrk_model_info(C.Structure):
    _fields_ = [("backend", C.c_int32), ("n_trees", C.c_int32), ("max_depth", C.c_int32), ("n_features", C.c_int32),
                ("is_f64", C.c_int32), ("n_categorical", C.c_int32), ("n_nodes", C.c_int64), ("n_leaves", C.c_int64),
                ("device_bytes", C.c_int64), ("base_score", C.c_double), ("bitvector", C.c_int32), ("tile_columns", C.c_int32)]


class mrk_encoder_info(C.Structure):
    _fields_ = [("layers", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("intermediate", C.c_int32),
                ("vocab", C.c_int32), ("max_positions", C.c_int32), ("type_vocab", C.c_int32), ("has_classifier", C.c_int32),
                ("max_length", C.c_int32), ("device_bytes", C.c_int64), ("layer_norm_eps", C.c_double)]


class mrk_field(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("n", C.c_int32), ("num", C.c_double), ("str", C.c_char_p),
                ("strs", C.POINTER(C.c_char_p)), ("nums", C.POINTER(C.c_double))]


class mrk_request(C.Structure):
    _fields_ = [("id", C.c_char_p), ("timestamp_ms", C.c_int64), ("user", C.c_char_p), ("session", C.c_char_p),
                ("fields", C.POINTER(mrk_field)), ("n_fields", C.c_int32), ("n_items", C.c_int32),
                ("item_ids", C.POINTER(C.c_char_p)), ("item_field_offsets", C.POINTER(C.c_int32)),
                ("item_fields", C.POINTER(mrk_field))]


class mrk_item_ids(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("offsets", C.c_void_p), ("bytes_len", C.c_size_t)]


# every symbol include/mrk.h declares: (restype, argtypes)
_V, _I, _P, _S = C.c_void_p, C.c_int, C.c_void_p, C.c_char_p
SIGNATURES = {
    "mrk_abi_version": (_I, []),
    "mrk_build_id": (_S, []),
    "mrk_config_kernel_keys": (_I, [_P, _S, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mrk_last_error": (_S, []),
    "mrk_init": (_I, [_P, _I, C.POINTER(_V)]),
    "mrk_device_count": (_I, []),
    "mrk_shutdown": (None, [_V]),
    "mrk_model_load": (_I, [_V, _I, _P, C.c_size_t, C.POINTER(_V)]),
    "mrk_model_load_container": (_I, [_V, _P, C.c_size_t, C.POINTER(_S), _I, C.POINTER(_V)]),
    "mrk_model_predict_f64": (_I, [_V, _P, _I, _I, _P]),
    "mrk_model_predict_device": (_I, [_V, _P, _I, _I, _P]),
    "mrk_model_get_info": (_I, [_V, C.POINTER(mrk_model_info)]),
    "mrk_model_inspect": (_I, [_I, _P, C.c_size_t, C.POINTER(mrk_model_info)]),
    "mrk_model_weights": (_I, [_V, _I, _P, _I]),
    "mrk_model_inspect_weights": (_I, [_I, _P, C.c_size_t, _I, _P, _I]),
    "mrk_abi_layout": (_I, [_P, _I]),
    "mrk_model_retain": (None, [_V]),
    "mrk_model_free": (None, [_V]),
    "mrk_config_load_json": (_I, [_V, _S, C.c_size_t]),
    "mrk_model_dim": (_I, [_V, _S]),
    "mrk_config_specialize": (_I, [_S, C.c_size_t, _S, _I, _I, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mrk_config_precompile": (_I, [_S, C.c_size_t, _S, _I, C.c_uint, _S, C.POINTER(C.c_int)]),
    "mrk_config_specialize_for_model": (_I, [_S, C.c_size_t, _S, _I, _P, C.c_size_t, _I, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mrk_config_precompile_for_model": (_I, [_S, C.c_size_t, _S, _I, _P, C.c_size_t, C.c_uint, _S, C.POINTER(C.c_int)]),
    "mrk_config_warmup": (_I, [_V, _S]),
    "mrk_store_put_double": (_I, [_V, _S, C.c_double]),
    "mrk_store_put_bool": (_I, [_V, _S, _I]),
    "mrk_store_put_string": (_I, [_V, _S, _S]),
    "mrk_store_put_string_list": (_I, [_V, _S, C.POINTER(_S), _I]),
    "mrk_store_put_double_list": (_I, [_V, _S, _P, _I]),
    "mrk_store_put_counter": (_I, [_V, _S, C.c_int64]),
    "mrk_store_put_periodic": (_I, [_V, _S, _P, _I]),
    "mrk_store_put_bounded_list": (_I, [_V, _S, C.POINTER(_S), _I]),
    "mrk_store_delete": (_I, [_V, _S]),
    "mrk_store_increment_periodic": (_I, [_V, _S, C.c_int64, C.c_int64]),
    "mrk_store_put_binary": (_I, [_V, _P, C.c_size_t, C.POINTER(C.c_int)]),
    "mrk_store_put_binary_at": (_I, [_V, _P, C.c_size_t, C.c_int64, C.POINTER(C.c_int)]),
    "mrk_store_expire": (_I, [_V, C.c_int64, C.POINTER(C.c_int64)]),
    "mrk_store_increment_periodic_batch": (_I, [_V, C.POINTER(_S), _P, _P, _I]),
    "mrk_store_increment": (_I, [_V, _S, C.c_int64]),
    "mrk_store_append": (_I, [_V, _S, _S, C.c_int64]),
    "mrk_store_flush": (_I, [_V]),
    "mrk_rank": (_I, [_V, _V, _S, C.POINTER(mrk_request), _P, _P, _P]),
    "mrk_rank_binary": (_I, [_V, _V, _S, _P, C.c_size_t, C.POINTER(C.c_int), _P, _P, _I]),
    "mrk_model_warmup": (_I, [_V, _V, _S, C.POINTER(C.c_int)]),
    "mrk_batch_prepare": (_I, [_V, _S, C.POINTER(mrk_request), _I, C.POINTER(_V)]),
    "mrk_batch_create": (_I, [_V, C.POINTER(_V)]),
    "mrk_batch_load": (_I, [_V, _S, C.POINTER(mrk_request), _I, _P]),
    "mrk_batch_enqueue_fetch": (_I, [_V]),
    "mrk_batch_host_outputs": (_I, [_V, C.POINTER(_V), C.POINTER(_V), C.POINTER(_V)]),
    "mrk_host_alloc": (_V, [C.c_size_t]),
    "mrk_host_free": (None, [_V]),
    "mrk_batch_total_items": (_I, [_V]),
    "mrk_batch_run": (_I, [_V, _V]),
    "mrk_batch_shard_chunk": (_I, [_V, _I]),
    "mrk_batch_run_shard": (_I, [_V, _V, _I, _I]),
    "mrk_shard_chunk": (C.c_int64, [C.c_int64, _I]),
    "mrk_shard_range": (_I, [C.c_int64, _I, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mrk_batch_sort": (_I, [_V]),
    "mrk_batch_stream": (_V, [_V]),
    "mrk_batch_sync": (_I, [_V]),
    "mrk_batch_device_outputs": (_I, [_V, C.POINTER(_V), C.POINTER(_V), C.POINTER(_V)]),
    "mrk_batch_fetch": (_I, [_V, _P, _P, _P]),
    "mrk_batch_status": (_I, [_V, _P]),
    "mrk_batch_free": (None, [_V]),
    "mrk_serve_start": (_I, [_V, _V, _S, _I, C.POINTER(_V)]),
    "mrk_serve_rank": (_I, [_V, C.POINTER(mrk_request), _P, _P]),
    "mrk_serve_stats": (_I, [_V, _P, _I]),
    "mrk_serve_stop": (None, [_V]),
    "mrk_comm_unique_id": (_I, [_P]),
    "mrk_comm_init": (_I, [_V, _P, _I, _I]),
    "mrk_comm_init_local": (_I, [C.POINTER(_V), _I]),
    "mrk_comm_rank": (_I, [_V]),
    "mrk_comm_world": (_I, [_V]),
    "mrk_comm_max_f64": (_I, [_V, C.POINTER(C.c_double)]),
    "mrk_comm_barrier": (_I, [_V]),
    "mrk_batch_run_sharded": (_I, [_V, _V]),
    "mrk_batch_allgather_scores": (_I, [_V]),
    "mrk_batch_gather_scores": (_I, [_V, C.POINTER(_V)]),
    "mrk_sync": (_I, [_V]),
    "mrk_stream": (_V, [_V]),
    "mrk_profile_enable": (_I, [_V, _I]),
    "mrk_profile_get": (_I, [_V, _S, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mrk_tokenizer_load": (_I, [_P, C.c_size_t, C.POINTER(_V)]),
    "mrk_tokenizer_encode_batch": (_I, [_V, C.POINTER(_S), C.POINTER(_S), _I, _P, _P, _P, _I, C.POINTER(C.c_int)]),
    "mrk_tokenizer_free": (None, [_V]),
    "mrk_encoder_load": (_I, [_V, _P, C.c_size_t, _P, C.c_size_t, _I, C.POINTER(_V)]),
    "mrk_encoder_load_ex": (_I, [_V, _P, C.c_size_t, _P, C.c_size_t, _I, _I, C.POINTER(_V)]),
    "mrk_checkpoint_describe": (_I, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mrk_encoder_get_info": (_I, [_V, C.POINTER(mrk_encoder_info)]),
    "mrk_encoder_embed": (_I, [_V, C.POINTER(_S), _I, _P]),
    "mrk_encoder_embed_ids": (_I, [_V, _P, _P, _P, _I, _I, _P]),
    "mrk_encoder_hidden_ids": (_I, [_V, _P, _P, _P, _I, _I, _P]),
    "mrk_encoder_score_pairs": (_I, [_V, C.POINTER(_S), C.POINTER(_S), _I, _P]),
    "mrk_encoder_score_ids": (_I, [_V, _P, _P, _P, _I, _I, _P]),
    "mrk_encoder_free": (None, [_V]),
    "mrk_config_bind_encoder": (_I, [_V, _S, _V]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libmrk_hip.so; fails loudly when the HIP extension is missing (no CPU fallback exists)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def reload_switches():
    """The library reads its experiment switches (MRK_* environment variables) once; tests and measurement
    scripts that change one inside a process call this afterwards."""
    fn = lib().mrk_debug_reload_switches
    fn.restype, fn.argtypes = None, []
    fn()


def check(status: int):
    if status != MRK_OK:
        msg = lib().mrk_last_error()
        raise MrkError(status, msg.decode("utf-8", "replace") if msg else "")
