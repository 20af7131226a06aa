"""Multi-GPU plumbing for the rank path: the path shards by request (or by the candidate items of one
big request) with a replicated store; the only exchange is one all-gather of the scores.

On the GPU the collective lives INSIDE the library (csrc/comm.cpp: RCCL linked directly, mrk_comm_* /
mrk_batch_run_sharded); what is left for the host is handing rank 0's 128-byte communicator id to the other ranks -
`exchange_unique_id` below does it with one TCP message per rank.  No torch anywhere in this package: the gloo restatement of
the merge for the CPU tests lives in tests/dist_helpers.py."""
from __future__ import annotations

import socket
import time


def exchange_unique_id(rank: int, world: int, make_id, addr: str = "127.0.0.1", port: int = 29500, timeout: float = 120.0) -> bytes:
    """Rank 0 calls make_id() (mrk_comm_unique_id), listens on addr:port and sends the id to the world - 1 ranks that
    connect; the others connect (retrying until rank 0 is up) and read it."""
    if world <= 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _peer = srv.accept()
            conn.sendall(uid)
            conn.close()
        srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    uid = b""
    while len(uid) < 128:
        chunk = c.recv(128 - len(uid))
        if not chunk:
            raise ConnectionError("rank 0 closed the connection before the communicator id arrived")
        uid += chunk
    c.close()
    return uid


def exchange_unique_id_file(rank: int, world: int, make_id, key: str, timeout: float = 120.0) -> bytes:
    """The same hand-over through a file in /tmp for the ranks of ONE node (bench.py under the driver's torchrun launch:
    the launcher's own store already listens on MASTER_PORT, so that port is not free for a second listener).  `key`
    must be the same on every rank of a launch and differ between launches (bench.py: parent pid + MASTER_PORT)."""
    import os

    if world <= 1:
        return make_id()
    path = f"/tmp/mrk_comm_{key}.id"
    if rank == 0:
        uid = make_id()
        tmp = path + f".{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)   # atomic: a reader sees nothing or all 128 bytes
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            if time.time() - os.path.getmtime(path) < 600:   # not a leftover of an old launch
                uid = open(path, "rb").read()
                if len(uid) == 128:
                    return uid
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError(f"no communicator id at {path}")
        time.sleep(0.02)


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous, balanced [lo, hi) chunk of n units for `rank` (first n % world ranks get one more)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def padded_chunk(n: int, world: int, tile: int = 128) -> int:
    """items per shard of an item-sharded run (== mrk_batch_shard_chunk): ceil(n / world) rounded up to
    whole scorer tiles, so every shard starts on a tile boundary and all shards have the same length"""
    per = -(-n // world)
    return -(-per // tile) * tile
