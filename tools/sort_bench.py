"""Ordering of ONE large request (csrc/bigsort.hip) by launch shape: HIP-event time of mrk_batch_sort for n candidates under a
few switch settings, same process, same box.   python tools/sort_bench.py [n ...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import metarank_amd as M  # noqa: E402
from backends import HipBackend  # noqa: E402
from workloads import ranklens  # noqa: E402

VARIANTS = [{}, {"MRK_BIG_SORT_BUCKET": "512"}, {"MRK_BIG_SORT_BUCKET": "256"}, {"MRK_BIG_SORT_BUCKET": "128"},
            {"MRK_BIG_SORT_BUCKET": "256", "MRK_BIG_SORT_TILE": "512"}, {"MRK_BIG_SORT_BUCKET": "2048"}]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100_000]
    hiprt = C.CDLL("libamdhip64.so")
    hiprt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = {}
    for n in sizes:
        scores = np.random.default_rng(n).normal(size=n)
        for env in VARIANTS:
            for k, v in env.items():
                os.environ[k] = v
            M.reload_switches()
            ctx = M.Context(0)
            hip = HipBackend(ranklens.ranklens_config(), "xgboost", ctx)
            reqs = [{"id": "r", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": [{"id": f"x{i}"} for i in range(n)]}]
            b = hip.ranker.new_batch()
            b.load("xgboost", M.RequestSet(reqs, pinned=False))
            b.run(None)
            b.sync()
            assert hiprt.hipMemcpy(b.device_outputs()[0], scores.ctypes.data, scores.nbytes, 1) == 0
            for _ in range(5):
                b.sort()
            b.sync()
            ctx.profile_enable(True)
            for _ in range(50):
                b.sort()
                b.sync()
            ms, k = ctx.profile_get("sort")
            ctx.profile_enable(False)
            import time
            t = time.perf_counter()
            for _ in range(50):
                b.sort()
            b.sync()
            wall = (time.perf_counter() - t) / 50 * 1e3
            out[f"{n} {env or 'default'}"] = {"event_ms": ms / max(k, 1), "back_to_back_ms": wall}
            print(n, env or "default", out[f"{n} {env or 'default'}"], flush=True)
            b.close()
            hip.close()
            ctx.close()
            for k_ in env:
                os.environ.pop(k_, None)
    M.reload_switches()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
