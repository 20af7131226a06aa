"""CPU: the host half of a rank batch (resolve_requests, csrc/features.cpp: ids -> slots, request constants, pre-pass table
sizes) gives the same device batch whatever the number of host threads - requests are resolved in parallel ranges and a
single large request spreads its id lookups.  No device: the harness (tools/host_bench.cpp) links the library's host code."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import host_bench


def test_threads_do_not_change_the_device_batch(tmp_path):
    exe = host_bench.build_exe(str(tmp_path / "host_bench"))
    dump = str(tmp_path / "small.txt")
    host_bench.write_dump(dump, "c2", catalogue=3000, sessions=300, n_req=400)  # 40 000 items: above the threading threshold
    sums = {t: host_bench.run(exe, dump, 400, t)[2] for t in (1, 3, 8)}
    assert len(set(sums.values())) == 1, sums


def test_one_large_request_spreads_its_lookups(tmp_path):
    exe = host_bench.build_exe(str(tmp_path / "host_bench"))
    dump = str(tmp_path / "one.txt")
    # ranklens.generate_requests draws with replacement when a request is larger than the catalogue
    import json
    from workloads import ranklens
    host_bench.write_dump(dump, "c2", catalogue=2000, sessions=50, n_req=1)
    lines = open(dump).read().split("\n")
    ev = ranklens.generate_requests(1, 20_000, 2000, 50)[0]
    lines = [l for l in lines if not l.startswith("R ")]
    lines.append("R " + "\t".join([ev["id"], ev["user"], ev["session"], str(ev["timestamp"])] + [it["id"] for it in ev["items"]]))
    open(dump, "w").write("\n".join(l for l in lines if l) + "\n")
    sums = {t: host_bench.run(exe, dump, 1, t)[2] for t in (1, 4)}
    assert len(set(sums.values())) == 1, sums
