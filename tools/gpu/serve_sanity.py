"""Staged sanity check of the serving queue's gangs (1 slot, 8 slots from Python threads, 64 slots): every result against
mrk_rank's, stage by stage, stopping at the first failure.  usage: python tools/gpu/serve_sanity.py [stage ...]"""
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import metarank_amd as M
from workloads import ranklens, synth

ctx = M.Context(0)
ranker = M.HipRanker(ranklens.ranklens_config(), ctx)
for kind, key, value in ranklens.generate_state(20_000, 2_000):
    getattr(ranker, "put_" + kind)(key, value)
ranker.flush()
events = ranklens.generate_requests(64, 100, 20_000, 2_000)
reqs = [M.Request(e) for e in events]
sample = ranker.prepare("xgboost", events[:32])
sample.run(None)
_, _, sm = sample.fetch(matrix=True)
sample.close()
booster = M.HipBooster(synth.synthetic_lgbm_model(n_trees=100, n_features=24, quantiles=ranklens.column_quantiles(sm)), M.LIGHTGBM, ctx)
want = [ranker.rerank("xgboost", r, booster)[1:] for r in reqs]
print("mrk_rank pass done", flush=True)
for n_slots, n_threads in ((1, 1), (8, 8), (64, 32)):
    t0 = time.time()
    srv = ranker.serve("xgboost", booster, n_slots=n_slots)
    print(f"slots {n_slots}: started in {time.time() - t0:.1f} s", flush=True)
    bad = [0]

    def client(t):
        for k in range(200):
            i = (t * 7 + k) % len(reqs)
            try:
                sc, od = srv.rerank(reqs[i])
            except Exception as e:
                print(f"thread {t} request {k}: {e}", flush=True)
                bad[0] += 1000
                return
            if not (np.array_equal(sc, want[i][0]) and np.array_equal(od, want[i][1])):
                bad[0] += 1
            if k % 50 == 49:
                time.sleep(0.03)   # longer than the workgroups' life: the gang is revived

    ts = [threading.Thread(target=client, args=(t,)) for t in range(n_threads)]
    t0 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    st = srv.stats()
    print(f"slots {n_slots} x {n_threads} threads: {bad[0]} differ, {time.time() - t0:.2f} s, queue {st['queue']} fallback {st['fallback']} launches {st['launches']}", flush=True)
    srv.close()
    if bad[0]:
        sys.exit(1)
print("ok")
