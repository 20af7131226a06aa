"""Closed-loop concurrent clients of mrk_rank (the reference's serving model: one rerank per request thread):
T threads x N sequential 100-item requests each.  Prints requests/s, items/s and the per-call p50 / p99 with
the batching front on (default) and off (MRK_RANK_COMBINE=0, set before the library loads)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import metarank_amd as M
from workloads import ranklens, synth

serve = "--serve" in sys.argv          # through the serving queue (mrk_serve_rank, one slot per thread) instead of mrk_rank
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
threads = int(argv[0]) if len(argv) > 0 else 16
per_thread = int(argv[1]) if len(argv) > 1 else 300
items = int(argv[2]) if len(argv) > 2 else 100
ctx = M.Context(0)
cfg = ranklens.ranklens_config()
ranker = M.HipRanker(cfg, ctx)
for kind, key, value in ranklens.generate_state(100_000, 10_000):
    getattr(ranker, "put_" + kind)(key, value)
ranker.flush()
events = ranklens.generate_requests(threads * 8, items, 100_000, 10_000)
sample = ranker.prepare("xgboost", events[:32])
sample.run(None)
_, _, sm = sample.fetch(matrix=True)
sample.close()
blob = synth.synthetic_lgbm_model(n_trees=500, n_features=24, quantiles=ranklens.column_quantiles(sm), cat_features=[7], cat_prob=0.007,
                                  missing="per_feature")
booster = M.HipBooster(blob, M.LIGHTGBM, ctx)
reqs = [M.Request(e) for e in events]
for r in reqs[:16]:
    ranker.rerank("xgboost", r, booster)
ranker.warmup_kernels("xgboost")
srv = ranker.serve("xgboost", booster, n_slots=min(threads, 64)) if serve else None
call = (lambda r: srv.rerank(r)) if serve else (lambda r: ranker.rerank("xgboost", r, booster))
for r in reqs[:16]:
    call(r)
lat = [[] for _ in range(threads)]
start = threading.Barrier(threads + 1)


def client(t):
    start.wait()
    for k in range(per_thread):
        r = reqs[(t * 8 + k) % len(reqs)]
        t0 = time.perf_counter()
        call(r)
        lat[t].append(time.perf_counter() - t0)


ts = [threading.Thread(target=client, args=(t,)) for t in range(threads)]
for t in ts:
    t.start()
start.wait()
t0 = time.perf_counter()
for t in ts:
    t.join()
wall = time.perf_counter() - t0
all_lat = np.concatenate([np.asarray(l) for l in lat]) * 1e3
n = threads * per_thread
print(f"{'serve queue' if serve else 'mrk_rank'} combine={os.environ.get('MRK_RANK_COMBINE', '1')} threads={threads} x {per_thread} requests of {items} items: "
      f"{n / wall:.0f} requests/s, {n * items / wall / 1e6:.2f} M items/s, per-call p50 {np.percentile(all_lat, 50):.3f} ms "
      f"p99 {np.percentile(all_lat, 99):.3f} ms")
if srv is not None:
    print("  ", srv.stats())
    srv.close()
