"""CPU: the host mirror of the device feature store (csrc/store.cpp) under heavy replacement - inline heaps, pools with
size-class recycling - against a reference map (tests/native/store_test.cpp).  store.cpp and features.cpp are compiled INTO
the test binary with AddressSanitizer + UBSan (a pool that hands out a range twice, an inline heap written past its record
end up as reports, not as silently wrong neighbours); the rest comes from libmrk_hip.so.  No device."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_store_mirror_matches_a_reference_and_stays_bounded(tmp_path):
    from metarank_amd import _native

    _native.build()
    exe = str(tmp_path / "store_test")
    lib_dir = os.path.dirname(_native.LIB_PATH)
    csrc = os.path.join(REPO, "metarank_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", os.path.join(REPO, "tests", "native", "store_test.cpp"), os.path.join(csrc, "store.cpp"),
                           os.path.join(csrc, "features.cpp"), "-I" + csrc, "-I" + os.path.join(REPO, "include"), "-L" + lib_dir, "-lmrk_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert " bad 0 bounded 1" in out.stdout
