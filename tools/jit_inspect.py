#!/usr/bin/env python
"""What the run-time specialised assembly kernels of a config look like, without a GPU: compiles them through the
host-only entry point mrk_config_specialize (hiprtc, gfx950) and prints each kernel's register / scratch / LDS use and
code size from the code object's metadata (llvm-readelf --notes).

    python tools/jit_inspect.py [c2|c3|c5] [--f32] [--model] [--kernel k] [--save out.co] [--asm out.s]
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metarank_amd import _native
from workloads import ranklens

LLVM = "/opt/rocm/lib/llvm/bin"


def specialise(cfg: dict, model: str, f64: bool, what: int) -> bytes:
    lib = _native.lib()
    blob = json.dumps({"features": cfg["features"], "models": cfg["models"]}).encode()
    need = C.c_size_t(0)
    lib.mrk_config_specialize(blob, len(blob), model.encode(), 1 if f64 else 0, what, None, 0, C.byref(need))
    buf = (C.c_uint8 * need.value)()
    _native.check(lib.mrk_config_specialize(blob, len(blob), model.encode(), 1 if f64 else 0, what, buf, need.value, C.byref(need)))
    return bytes(buf[:need.value])


def bench_like_model(cfg: dict, wl: str) -> bytes:
    """A forest shaped like bench.py's (500 leaf-wise 16-leaf trees, thresholds at the columns' quantiles, one missing type per
    column, a categorical column): the sample matrix comes from the CPU oracle over a small generated state."""
    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from backends import OracleBackend
    from workloads import synth

    orc = OracleBackend(cfg, "xgboost")
    ranklens.load_state(orc, ranklens.generate_state(2000, 200, c3=(wl == "c3")))
    sample = np.concatenate([orc.matrix(ev) for ev in ranklens.generate_requests(16, 100, 2000, 200, seed=ranklens.SEED + 99)])
    return synth.synthetic_lgbm_model(n_trees=500, n_features=sample.shape[1], num_leaves=16, max_depth=8, quantiles=ranklens.column_quantiles(sample),
                                      cat_features=[7], cat_prob=0.007, missing="per_feature")


def specialise_for_model(cfg: dict, model: str, blob_model: bytes, what: int) -> bytes:
    lib = _native.lib()
    blob = json.dumps({"features": cfg["features"], "models": cfg["models"]}).encode()
    need = C.c_size_t(0)
    lib.mrk_config_specialize_for_model(blob, len(blob), model.encode(), 0, blob_model, len(blob_model), what, None, 0, C.byref(need))
    buf = (C.c_uint8 * need.value)()
    _native.check(lib.mrk_config_specialize_for_model(blob, len(blob), model.encode(), 0, blob_model, len(blob_model), what, buf, need.value, C.byref(need)))
    return bytes(buf[:need.value])


def main():
    wl = next((a for a in sys.argv[1:] if not a.startswith("-")), "c2")
    f64 = "--f32" not in sys.argv
    cfg = {"c2": ranklens.ranklens_config, "c3": ranklens.c3_config, "c5": ranklens.c5_config}[wl]()
    what = 1
    if "--kernel" in sys.argv:   # one kernel's translation unit (0 the workgroup-per-request kernel, ... as in jit.hpp)
        what |= (int(sys.argv[sys.argv.index("--kernel") + 1]) + 1) << 8
    if "--model" in sys.argv:    # keyed by the view signature of a bench-like LightGBM forest too (the kernels the benchmark runs)
        code = specialise_for_model(cfg, "xgboost", bench_like_model(cfg, wl), what)
    else:
        code = specialise(cfg, "xgboost", f64, what)
    path = "/tmp/mrk_jit_inspect.co"
    if "--save" in sys.argv:
        path = sys.argv[sys.argv.index("--save") + 1]
    open(path, "wb").write(code)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        agpr = re.match(r":?\s*(\d+)", blk)
        print("%-28s vgpr %4s agpr %3s sgpr %4s scratch %5s B  lds %6s B  spills v%s s%s" % (
            g("name"), g("vgpr_count"), agpr.group(1) if agpr else "?", g("sgpr_count"), g("private_segment_fixed_size"),
            g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", path], capture_output=True, text=True).stdout
    for ln in syms.splitlines():
        if "FUNC" in ln and "mrk_jit" in ln:
            f = ln.split()
            print(f"{f[-1]:28s} code {int(f[2])} bytes")
    if "--asm" in sys.argv:
        out = sys.argv[sys.argv.index("--asm") + 1]
        subprocess.run([f"{LLVM}/llvm-objdump", "-d", path], stdout=open(out, "w"), check=True)
        print("disassembly ->", out)


if __name__ == "__main__":
    main()
