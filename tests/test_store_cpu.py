"""CPU: the host mirror of the device feature store (csrc/store.cpp) under heavy replacement - inline heaps, pools with
size-class recycling - against a reference map (tests/native/store_test.cpp).  Links libmrk_hip.so; no device."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_store_mirror_matches_a_reference_and_stays_bounded(tmp_path):
    from metarank_amd import _native

    _native.build()
    exe = str(tmp_path / "store_test")
    lib_dir = os.path.dirname(_native.LIB_PATH)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                           os.path.join(REPO, "tests", "native", "store_test.cpp"), "-o", exe,
                           "-I" + os.path.join(REPO, "metarank_amd", "csrc"), "-L" + lib_dir, "-lmrk_hip", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " bad 0 bounded 1" in out.stdout
