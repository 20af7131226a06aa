"""Micro-benchmark of the write path: N PeriodicIncrements staged through the C ABI, then one flush
(host grouping + upload + bucket-ring kernel + window sums).  Prints increments/s for both halves."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import metarank_amd as M
from metarank_amd import _native as N
from workloads import ranklens

n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_inc = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = M.Context(0)
ranker = M.HipRanker(ranklens.ranklens_config(), ctx)
rng = np.random.default_rng(0)
items = rng.integers(n_items, size=n_inc)
kinds = rng.integers(2, size=n_inc)
keys = [(b"item=%d/ctr_click" % i) if k else (b"item=%d/ctr_impression" % i) for i, k in zip(items, kinds)]
arr = (C.c_char_p * n_inc)(*keys)
ts = (ranklens.TS + rng.integers(-30 * 86_400_000, 86_400_000, size=n_inc)).astype(np.int64)
inc = np.ones(n_inc, dtype=np.int64)
L = N.lib()
for rnd in range(3):
    t0 = time.perf_counter()
    N.check(L.mrk_store_increment_periodic_batch(ctx.handle, arr, ts.ctypes.data_as(C.c_void_p), inc.ctypes.data_as(C.c_void_p), n_inc))
    t1 = time.perf_counter()
    ranker.flush()
    t2 = time.perf_counter()
    print(f"round {rnd}: stage {n_inc / (t1 - t0) / 1e6:.2f} M inc/s (host: key parse + slot lookup), "
          f"flush {n_inc / (t2 - t1) / 1e6:.2f} M inc/s (sort/group + upload + ring kernel + window sums) "
          f"[{(t2 - t1) * 1e3:.1f} ms for {n_inc} increments on {n_items} items]")
