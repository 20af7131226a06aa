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
