"""CPU (gloo, world size 2) cover of the N > 1 path: shard ranges and the score all-gather merge."""
import os
import subprocess
import sys

from metarank_amd.dist import shard_range

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 100, 100_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_library_shard_arithmetic_equals_the_python_restatement():
    """mrk_shard_chunk / mrk_shard_range (host-only entry points; what mrk_batch_shard_chunk and mrk_batch_run_shard use)
    against dist.padded_chunk: every n in 0..4100, a spread up to 10^6 and the sizes of the benchmark, world 1..8 - plus
    the partition properties the all-gather relies on (tile-aligned equal chunks that cover [0, n) exactly once)."""
    import ctypes as C

    from metarank_amd import _native as N
    from metarank_amd.dist import padded_chunk

    L = N.lib()
    ns = list(range(0, 4101)) + list(range(4101, 1_000_001, 9973)) + [100_000, 384_000, 1_000_000, 4_000_000, (1 << 31) - 1]
    lo, hi = C.c_int64(), C.c_int64()
    for world in range(1, 9):
        for n in ns:
            chunk = L.mrk_shard_chunk(n, world)
            assert chunk == padded_chunk(n, world), (n, world)
            assert chunk % 128 == 0 and chunk * world >= n
            if n % 97 == 0 or n < 300:
                end = 0
                for r in range(world):
                    assert L.mrk_shard_range(n, r, world, C.byref(lo), C.byref(hi)) == N.MRK_OK
                    assert lo.value == min(r * chunk, n) == end and hi.value == min((r + 1) * chunk, n)
                    end = hi.value
                assert end == n
    assert L.mrk_shard_chunk(-1, 2) < 0 and L.mrk_shard_chunk(10, 0) < 0
    assert L.mrk_shard_range(10, 2, 2, C.byref(lo), C.byref(hi)) == N.ERR_INVALID_ARG


def test_bench_start_up_with_two_ranks_and_no_device():
    """bench.py --gpus 2 --workload c4 under the driver's launcher, as far as it goes without a GPU (MRK_BENCH_DRY_DIST):
    both ranks must end up with the same communicator id, mrk_comm_init must reach its argument checks, and the two
    shard ranges must tile the request."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MRK_BENCH_DRY_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "c4"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    rows = sorted((json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{"dry"')), key=lambda d: d["rank"])
    assert [d["rank"] for d in rows] == [0, 1] and rows[0]["uid_crc"] == rows[1]["uid_crc"]
    assert rows[0]["lo"] == 0 and rows[0]["hi"] == rows[1]["lo"] == rows[0]["chunk"] == 50048 and rows[1]["hi"] == 100_000


def test_two_rank_gloo_merge():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(REPO, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_OK 2" in p.stdout


def test_communicator_id_hand_over_without_torch(tmp_path):
    """bench.py --gpus N hands rank 0's 128-byte ncclUniqueId to the other ranks through a file (single node, under the
    launcher) - or one TCP message per rank: both with plain processes, no torch, no GPU."""
    import multiprocessing as mp

    from metarank_amd.dist import exchange_unique_id, exchange_unique_id_file

    uid = bytes(range(128))
    key = f"test_{os.getpid()}"
    port = 29731 + os.getpid() % 2000

    def worker(rank, q, how):
        if how == "file":
            q.put((rank, exchange_unique_id_file(rank, 3, lambda: uid, key, timeout=30)))
        else:
            q.put((rank, exchange_unique_id(rank, 3, lambda: uid, "127.0.0.1", port, timeout=30)))

    for how in ("file", "tcp"):
        q = mp.Queue()
        ps = [mp.Process(target=worker, args=(r, q, how)) for r in (1, 2, 0)]   # rank 0 last: the others wait for it
        for p in ps:
            p.start()
        got = dict(q.get(timeout=60) for _ in ps)
        for p in ps:
            p.join(30)
        assert got == {0: uid, 1: uid, 2: uid}, how
    os.remove(f"/tmp/mrk_comm_{key}.id")
