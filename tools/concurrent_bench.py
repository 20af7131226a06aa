"""Closed-loop concurrent clients of the per-request entry points (the reference's serving model: one Ranker.rerank per request
fiber, api/routes/RankApi.scala:25-41): T NATIVE threads x N sequential requests each through mrk_rank's batching front, and -
with --serve - through the serving queue (mrk_serve_rank).  The threads are C++ (tools/native/callers_driver.cpp, against
include/mrk.h only); every concurrent result is compared bit for bit with a sequential pass over the same requests.

    python tools/concurrent_bench.py [--serve | --queue] [--json] [--lanes N] [threads,threads,... [requests_per_thread [items]]]

--queue: the callers use mrk_rank while a serving queue of the model is started (mrk_serve_start once, as a host does at warm-up):
the library answers them through the queue's resident workgroups and sends the overflow through the batching front.

Prints one line per thread count: requests/s, items/s, per-call p50 / p99 (R-6 percentiles) and the number of results that
differed from the sequential pass (must be 0).  (Until round 5 this tool drove Python threads: at 16+ callers it measured the
interpreter lock - every call re-entered Python - not the library; `--python` keeps that loop for comparison.)"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
flags = [a for a in sys.argv[1:] if a.startswith("--")]
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--lanes" in flags:   # (must be set before the library reads its switches)
    os.environ["MRK_RANK_LANES"] = argv.pop(0)
slots_wish = int(argv.pop(0)) if "--slots" in flags else 64
import metarank_amd as M
from metarank_amd import _native as N
from metarank_amd.request import request_array
from workloads import ranklens, synth

serve = "--serve" in flags
queue = "--queue" in flags   # mrk_rank while a serving queue of this model is started: the library routes the calls through it
thread_counts = [int(x) for x in (argv[0] if argv else "1,4,16,32,64,128,256").split(",")]
per_thread = int(argv[1]) if len(argv) > 1 else 400
items = int(argv[2]) if len(argv) > 2 else 100


def driver():
    src = os.path.join(REPO, "tools", "native", "callers_driver.cpp")
    so = os.path.join(REPO, "tools", "native", "libcallers_driver.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-I", os.path.join(REPO, "include"), src, "-o", so,
                               "-L", os.path.dirname(N.LIB_PATH), "-lmrk_hip", "-Wl,-rpath," + os.path.dirname(N.LIB_PATH), "-pthread"])
    N.lib()   # libmrk_hip.so is in the process before the driver resolves it
    d = C.CDLL(so)
    d.mrk_bench_callers.restype = C.c_int
    d.mrk_bench_callers.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return d


def setup(n_requests):
    ctx = M.Context(0)
    cfg = ranklens.ranklens_config()
    ranker = M.HipRanker(cfg, ctx)
    for kind, key, value in ranklens.generate_state(100_000, 10_000):
        getattr(ranker, "put_" + kind)(key, value)
    ranker.flush()
    events = ranklens.generate_requests(n_requests, items, 100_000, 10_000)
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
    return ctx, ranker, booster, reqs


def sequential_results(ranker, booster, reqs):
    sc = np.zeros((len(reqs), items), dtype=np.float64)
    od = np.zeros((len(reqs), items), dtype=np.int32)
    for i, r in enumerate(reqs):
        _, s, o = ranker.rerank("xgboost", r, booster)
        sc[i, :len(s)] = s
        od[i, :len(o)] = o
    return sc, od


def run_native(d, ctx, booster, srv_handle, arr, n_reqs, threads, sc, od):
    lat = np.zeros(threads * per_thread, dtype=np.float64)
    out = np.zeros(8, dtype=np.float64)
    rc = d.mrk_bench_callers(ctx.handle, booster.handle, b"xgboost", srv_handle, C.addressof(arr), n_reqs, items, threads, per_thread,
                             lat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    n = threads * per_thread
    return {"callers": threads, "requests_per_s": round(n / out[0]), "items_per_s": round(n * items / out[0]),
            "p50_ms": round(float(np.percentile(lat, 50, method="weibull")), 4), "p99_ms": round(float(np.percentile(lat, 99, method="weibull")), 4),
            "max_ms": round(float(lat.max()), 3), "first_error": int(out[1]), "results_differing_from_sequential": int(out[3])}


def run_python(ranker, booster, srv, reqs, threads):
    call = (lambda r: srv.rerank(r)) if srv is not None else (lambda r: ranker.rerank("xgboost", r, booster))
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
    return {"callers": threads, "requests_per_s": round(n / wall), "items_per_s": round(n * items / wall),
            "p50_ms": round(float(np.percentile(all_lat, 50)), 4), "p99_ms": round(float(np.percentile(all_lat, 99)), 4), "driver": "python threads"}


def main():
    n_requests = max(thread_counts) * 8
    ctx, ranker, booster, reqs = setup(n_requests)
    sc, od = sequential_results(ranker, booster, reqs)
    arr = request_array(reqs)
    d = driver()
    rows = []
    for threads in thread_counts:
        srv = ranker.serve("xgboost", booster, n_slots=min(threads, slots_wish)) if (serve or queue) else None
        if srv is not None:
            for r in reqs[:16]:
                srv.rerank(r)
        if "--python" in flags:
            row = run_python(ranker, booster, srv, reqs, threads)
        else:
            h = srv._h if (srv is not None and serve) else None
            # warm: lanes, streams, pinned buffers.  With a queue started the callers beyond its slots are the front's: a warm-up of 8
            # callers never reaches it, and the front's scratch batches would find their sizes (allocations: tens of ms each) inside
            # the measured window
            run_native(d, ctx, booster, h, arr, len(reqs), threads if ('--warm-full' in flags or srv is not None) else min(threads, 8), sc, od)
            row = run_native(d, ctx, booster, h, arr, len(reqs), threads, sc, od)
        row["path"] = "mrk_serve_rank" if serve else ("mrk_rank+queue" if queue else "mrk_rank")
        if srv is not None:
            row["serve_stats"] = srv.stats()
            srv.close()
        if os.environ.get("MRK_FRONT_TRACE"):
            L = N.lib()
            buf = np.zeros((200000, 5), dtype=np.float32)
            L.mrk_debug_front_trace.restype = C.c_int
            nrow = min(L.mrk_debug_front_trace(buf.ctypes.data_as(C.c_void_p), len(buf)), len(buf))
            tr = buf[:nrow]
            if nrow:
                tot = tr[:, 1:].sum(axis=1)
                worst = np.argsort(tot)[-8:]
                print(f"   front trace: {nrow} batches, mean size {tr[:, 0].mean():.1f}, mean us build {tr[:, 1].mean():.0f} run {tr[:, 2].mean():.0f} fetch {tr[:, 3].mean():.0f} copy {tr[:, 4].mean():.0f}; "
                      f"p99 us build {np.percentile(tr[:, 1], 99):.0f} run {np.percentile(tr[:, 2], 99):.0f} fetch {np.percentile(tr[:, 3], 99):.0f}")
                for w in worst:
                    print("      slow batch: n=%d build %.0f run %.0f fetch %.0f copy %.0f us" % tuple(tr[w]))
        rows.append(row)
        if "--json" not in flags:
            print(f"{row['path']} lanes={os.environ.get('MRK_RANK_LANES', 'default')} callers={threads} x {per_thread} requests of {items} items: "
                  f"{row['requests_per_s']} requests/s, {row['items_per_s'] / 1e6:.2f} M items/s, per-call p50 {row['p50_ms']:.3f} ms p99 {row['p99_ms']:.3f} ms"
                  + (f" max {row['max_ms']:.1f} ms, {row['results_differing_from_sequential']} results differ, first error {row['first_error']}" if "max_ms" in row else " (python threads)"),
                  flush=True)
    if "--json" in flags:
        print(json.dumps(rows))
    return rows


if __name__ == "__main__":
    main()
