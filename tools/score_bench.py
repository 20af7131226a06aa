"""Micro-benchmark of the scorer alone (device-resident matrix): items/s and node visits/s."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import metarank_amd as M
from metarank_amd import _native as N
from workloads import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 409600
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kind = sys.argv[3] if len(sys.argv) > 3 else "lgbm"
ntrees = int(sys.argv[4]) if len(sys.argv) > 4 else 500
rng = np.random.default_rng(0)
X = rng.normal(size=(rows, cols))
q = [np.quantile(X[:, j], np.linspace(0.02, 0.98, 49)) for j in range(cols)]
ctx = M.Context(0)
if kind == "lgbm":
    b = M.HipBooster(synth.synthetic_lgbm_model(n_trees=ntrees, n_features=cols, quantiles=q, missing=os.environ.get("MISSING", "per_feature"),
                                                cat_features=[7], cat_prob=0.007), M.LIGHTGBM, ctx)
else:
    b = M.HipBooster(synth.synthetic_xgb_model(n_trees=ntrees, n_features=cols, depth=6, quantiles=q), M.XGBOOST, ctx)
print(b.info())
hip = C.CDLL("libamdhip64.so")
dx, dout = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(dx), C.c_size_t(X.nbytes)) == 0
assert hip.hipMalloc(C.byref(dout), C.c_size_t(rows * 8)) == 0
assert hip.hipMemcpy(dx, X.ctypes.data_as(C.c_void_p), C.c_size_t(X.nbytes), 1) == 0
L = N.lib()
for _ in range(3):
    N.check(L.mrk_model_predict_device(b.handle, dx, rows, cols, dout))
ctx.sync()
ctx.profile_enable(True)
K = 20
t0 = time.perf_counter()
for _ in range(K):
    N.check(L.mrk_model_predict_device(b.handle, dx, rows, cols, dout))
ctx.sync()
dt = (time.perf_counter() - t0) / K
ms, n = ctx.profile_get("score")
bms, bn = ctx.profile_get("bin")
print(f"rows={rows} cols={cols} {kind} trees={ntrees} R={os.environ.get('MRK_QS_R')} scorer={os.environ.get('MRK_SCORER')}: "
      f"wall {dt*1e3:.3f} ms/launch, score {ms/n:.3f} ms, bin {bms/max(bn,1):.3f} ms, {rows/dt/1e6:.1f} M items/s")
