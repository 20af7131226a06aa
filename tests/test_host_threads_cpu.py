"""CPU: the parallel host stage of the pointer-style prepare (csrc/features.cpp resolve_requests over MRK_HOST_THREADS
ranges of requests) under ThreadSanitizer: no data race, and the bytes the device would receive do not depend on the
thread count.  store.cpp and features.cpp are compiled into the harness (tools/host_bench.cpp) with -fsanitize=thread."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parallel_resolve_has_no_races_and_one_result(tmp_path):
    from metarank_amd import _native

    sys.path.insert(0, os.path.join(REPO, "tools"))
    import host_bench

    _native.build()
    lib_dir = os.path.dirname(_native.LIB_PATH)
    csrc = os.path.join(REPO, "metarank_amd", "csrc")
    exe = str(tmp_path / "host_bench_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(REPO, "tools", "host_bench.cpp"), os.path.join(csrc, "store.cpp"), os.path.join(csrc, "features.cpp"),
                           "-I" + csrc, "-I" + os.path.join(REPO, "include"), "-L" + lib_dir, "-lmrk_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe])
    dump = str(tmp_path / "dump.txt")
    host_bench.write_dump(dump, "c2", catalogue=3000, sessions=300, n_req=512)   # 51 200 items: above the parallel threshold
    sums = []
    for threads in ("1", "8"):
        out = subprocess.run([exe, dump, "512", threads], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
        text = out.stdout + out.stderr
        assert out.returncode == 0 and "ThreadSanitizer" not in text, text[-4000:]
        sums.append(re.search(r"checksum ([0-9a-f]+)", text).group(1))
    assert sums[0] == sums[1]
