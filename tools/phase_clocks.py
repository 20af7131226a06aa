#!/usr/bin/env python
"""Where the fused assembly kernel spends its time (measurement build: MRK_DEFINES=MRK_PHASE_CLOCKS python -c 'from
metarank_amd import _native; _native.build(force=True)').  Runs the c2 batch a few times and prints the core-clock
cycles thread 0 of a workgroup spends per phase and per op (lane 0), averaged over workgroups."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metarank_amd as M
from metarank_amd import _native
from workloads import ranklens, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n_req, n_items = (3840, 100) if wl == "c2" else (384, 1000)
if len(sys.argv) > 2:
    n_req = int(sys.argv[2])   # few requests: every workgroup alone on its CU - the cycles are the critical path itself
ctx = M.Context(0)
cfg = ranklens.c3_config() if wl == "c3" else ranklens.ranklens_config()
ranker = M.HipRanker(cfg, ctx)
ranklens.load_state(ranker, ranklens.generate_state(100_000, 10_000, c3=(wl == "c3")))
ranker.flush()
dim = ranker.dim("xgboost")
sample = ranker.prepare("xgboost", ranklens.generate_requests(64, 100, 100_000, 10_000, seed=ranklens.SEED + 99))
sample.run(None)
_, _, sm = sample.fetch(matrix=True)
sample.close()
blob = synth.synthetic_lgbm_model(n_trees=500, n_features=dim, num_leaves=16, max_depth=8, quantiles=ranklens.column_quantiles(sm),
                                  cat_features=[7], cat_prob=0.007, missing="per_feature")
booster = M.HipBooster(blob, M.LIGHTGBM, ctx)
batch = ranker.prepare("xgboost", ranklens.generate_requests(n_req, n_items, 100_000, 10_000))
lib = _native.lib()
lib.mrk_debug_phase.restype = C.c_int
lib.mrk_debug_phase.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
out = (C.c_uint64 * 64)()
for _ in range(3):
    batch.run(booster)
batch.sync()
rc = lib.mrk_debug_phase(ctx._h, b"xgboost", out)
assert rc == 0, f"mrk_debug_phase -> {rc}"
reps = 5
for _ in range(reps):
    batch.run(booster)
batch.sync()
assert lib.mrk_debug_phase(ctx._h, b"xgboost", out) == 0
v = np.array(list(out), dtype=np.float64)
wg = v[6]
# (round 6, wave-per-section pre-pass: [1] = both sections + their barrier, [2] / [3] = the interacted_with section on wavefront 1 and
#  the diversity section on wavefront 0, each by itself; [7] = issuing the op groups' primary cells and second trips, part of [5])
names = ["table sweep", "pre-pass sections + barrier", "  interacted_with alone (wave 1)", "  diversity alone (wave 0)", "diversity medians (workgroup-wide path)", "per-item assembly"]
print(f"workgroups {int(wg)}; cycles per workgroup (thread 0):")
tot = (v[0] + v[1] + v[5]) / wg
for i, n in enumerate(names):
    print(f"  {n:28s} {v[i] / wg:10.0f}  {100 * v[i] / wg / tot:5.1f} %")
print(f"  {'  of which issuing trips':28s} {v[7] / wg:10.0f}")
print(f"  diversity section: find-first {v[8] / wg:.0f}, strings {v[9] / wg:.0f}, numeric medians {v[10] / wg:.0f}")
print(f"  {'total':28s} {tot:10.0f}")
model_feats = cfg["models"]["xgboost"]["features"]
ops = v[16:16 + len(model_feats)]
print("per op (lane 0 of wave 0, first round of items):")
for n, c in zip(model_feats, ops):
    print(f"  {n:28s} {c / wg:10.0f}")
