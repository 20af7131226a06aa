"""Worker for tests/test_dist_cpu.py: launched as 2 gloo ranks.  No GPU: the per-shard compute is
the CPU oracle; what is under test is the sharding + all-gather merge used by bench.py --gpus N."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from backends import OracleBackend  # noqa: E402
from workloads import ranklens, synth  # noqa: E402
from metarank_amd.dist import padded_chunk, shard_range  # noqa: E402
from dist_helpers import all_gather_padded, all_gather_scores  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = ranklens.ranklens_config()
    b = OracleBackend(cfg, "xgboost")
    ranklens.load_state(b, ranklens.generate_state(600, 60))  # replicated store
    reqs = ranklens.generate_requests(7, 50, 600, 60, seed=9)  # 7 requests: uneven shards
    blob = synth.synthetic_lgbm_model(n_trees=40, n_features=24, seed=4)
    b.load_model(blob, 0)
    # (1) request sharding: every rank ranks its own requests, scores are merged
    lo, hi = shard_range(len(reqs), rank, world)
    local = np.concatenate([b.rerank(ev)[1] for ev in reqs[lo:hi]]) if hi > lo else np.zeros(0)
    # requests have equal sizes here, so item counts follow request counts
    merged = all_gather_scores(torch.from_numpy(local))  # uneven shards: sizes are exchanged first
    # (2) item sharding of ONE request (C4): assemble once (request-level reductions need all items),
    #     score a contiguous chunk of rows per rank, merge
    big = ranklens.generate_requests(1, 333, 600, 60, seed=10)[0]
    m = b.matrix(big)
    lo2, hi2 = shard_range(len(m), rank, world)
    part = b.forest.predict(m[lo2:hi2])
    merged2 = all_gather_scores(torch.from_numpy(part), [shard_range(len(m), r, world)[1] - shard_range(len(m), r, world)[0] for r in range(world)])
    # (3) the same the way the library shards it (mrk_batch_run_shard): equal tile-aligned chunks, each rank
    #     fills its slice of one padded buffer, one in-place all-gather
    #     - the chunk and this rank's range come from the LIBRARY (mrk_shard_chunk / mrk_shard_range, host-only entry points)
    import ctypes as C

    from metarank_amd import _native as N

    L = N.lib()
    chunk = int(L.mrk_shard_chunk(len(m), world))
    assert chunk == padded_chunk(len(m), world)
    c_lo, c_hi = C.c_int64(), C.c_int64()
    assert L.mrk_shard_range(len(m), rank, world, C.byref(c_lo), C.byref(c_hi)) == N.MRK_OK
    buf = torch.zeros(chunk * world, dtype=torch.float64)
    lo3, hi3 = c_lo.value, c_hi.value
    buf[lo3:hi3] = torch.from_numpy(b.forest.predict(m[lo3:hi3]))
    all_gather_padded(buf, chunk)
    assert chunk % 128 == 0 and np.array_equal(buf[:len(m)].numpy(), b.forest.predict(m))
    # ... and the ORDER every rank derives from the gathered scores (mrk_batch_run_sharded: all-gather, then every rank sorts) is the
    # unsharded request's: sortBy(-score), stable, java.lang.Double.compare (ml/Ranker.scala:52-67) - on EVERY rank, ties included
    _, full_scores, full_order = b.rerank(big)
    gathered = buf[:len(m)].numpy()
    bits = (-gathered).view(np.uint64).copy()               # Double.compare order of -score: sign-magnitude bits -> unsigned key
    bits[np.isnan(gathered)] = 0x7FF8000000000000
    neg = (bits >> np.uint64(63)) != 0
    key = np.where(neg, ~bits, bits | np.uint64(1 << 63))
    order = np.argsort(key, kind="stable")
    assert np.array_equal(gathered, full_scores) and order.tolist() == full_order.tolist(), f"rank {rank}: merged order differs from the unsharded oracle"
    if rank == 0:
        full = np.concatenate([b.rerank(ev)[1] for ev in reqs])
        assert np.array_equal(merged.numpy(), full)
        assert np.array_equal(merged2.numpy(), b.forest.predict(m))
        print("DIST_OK", world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
