#!/usr/bin/env python
"""Host side of a rank batch, without a GPU: how fast does the library turn RankingEvents (string ids) into the device
batch (slots, request constants, pre-pass table sizes)?  That is resolve_requests() of csrc/features.cpp - everything
mrk_batch_prepare does before the upload.  Writes the synthetic Ranklens state and requests to a text file and runs
tools/host_bench.cpp (compiled here against libmrk_hip.so) on it.

    python tools/host_bench.py [c2|c3] [n_requests] [threads]
"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from metarank_amd import _native
from workloads import ranklens


def write_dump(path, wl="c2", catalogue=100_000, sessions=10_000, n_req=7680):
    """config + state puts + requests as tab-separated lines (C / P / R records) for tools/host_bench.cpp"""
    n_items = 1000 if wl == "c3" else 100
    cfg = ranklens.c3_config() if wl == "c3" else ranklens.ranklens_config()
    with open(path + ".tmp", "w") as f:
        f.write("C " + json.dumps(cfg) + "\n")
        for kind, key, v in ranklens.generate_state(catalogue, sessions, c3=(wl == "c3")):
            vals = [v] if kind in ("double", "string", "counter") else list(v)
            f.write("P " + kind + "\t" + key + "\t" + "\t".join(repr(float(x)) if kind in ("double", "double_list") else str(x) for x in vals) + "\n")
        for ev in ranklens.generate_requests(n_req, min(n_items, catalogue), catalogue, sessions):
            f.write("R " + "\t".join([ev["id"], ev["user"], ev["session"], str(ev["timestamp"])] + [it["id"] for it in ev["items"]]) + "\n")
    os.rename(path + ".tmp", path)


def build_exe(exe="/tmp/mrk_host_bench"):
    _native.build()
    src = os.path.join(REPO, "tools", "host_bench.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(_native.LIB_PATH)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", exe,
                               "-I" + os.path.join(REPO, "metarank_amd", "csrc"), "-L" + os.path.dirname(_native.LIB_PATH), "-lmrk_hip",
                               "-Wl,-rpath," + os.path.dirname(_native.LIB_PATH), "-lpthread"])
    return exe


def run(exe, dump, n_req, threads):
    """-> (ms per batch, items/s, checksum of what the device would receive)"""
    out = subprocess.run([exe, dump, str(n_req), str(threads)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"resolve_requests: ([\d.]+) ms .* -> ([\d.]+) M items/s", out)
    c = re.search(r"checksum ([0-9a-f]+)", out)
    return float(m.group(1)), float(m.group(2)) * 1e6, c.group(1), out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    threads = sys.argv[3] if len(sys.argv) > 3 else "1"
    dump = f"/tmp/mrk_host_bench_{wl}.txt"
    if not os.path.exists(dump):
        write_dump(dump, wl, n_req=max(n_req, 7680))
    print(run(build_exe(), dump, n_req, threads)[3], end="")
