#!/bin/bash
# builds the current tree's library + the stock program's two batch kernels into ab/<name>/ (same-box A/B runs: MRK_LIB=ab/<name>/libmrk_hip.so)
set -e
name=$1
cd "$(dirname "$0")/.."
python -c "
from metarank_amd import _native
_native.build()
" 2>&1 | grep -E " error" -A5 || true
mkdir -p ab/$name/jit_cache
rm -f ab/$name/jit_cache/*.co
cp metarank_amd/libmrk_hip.so ab/$name/
MRK_LIB=$PWD/ab/$name/libmrk_hip.so python - <<PY
import ctypes as C, json, os
from concurrent.futures import ThreadPoolExecutor
from metarank_amd import _native
from workloads import ranklens
L = _native.lib()
out = os.path.join("ab", "$name", "jit_cache")
def one(args):
    cfg, k = args
    blob = json.dumps({"features": cfg["features"], "models": cfg["models"]}).encode()
    n = C.c_int(0)
    _native.check(L.mrk_config_precompile(blob, len(blob), b"xgboost", 1, 1 << k, out.encode(), C.byref(n)))
with ThreadPoolExecutor(4) as ex:
    list(ex.map(one, [(ranklens.ranklens_config(), 0), (ranklens.ranklens_config(), 1), (ranklens.c3_config(), 1), (ranklens.ranklens_config(), 4)]))
print("$name", sorted(os.listdir(out)))
PY
