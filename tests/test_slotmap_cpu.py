"""CPU: SlotMap (csrc/store.hpp), the id -> slot map of the device feature store's host side, against std::unordered_map:
present ids resolve to their slot, absent ones (near misses included) never resolve.  Native test, compiled here with AddressSanitizer + UBSan."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slotmap_matches_a_reference_map(tmp_path):
    exe = str(tmp_path / "slotmap_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", os.path.join(REPO, "tests", "native", "slotmap_test.cpp"), "-o", exe,
                           "-I" + os.path.join(REPO, "metarank_amd", "csrc"), "-I" + os.path.join(REPO, "include"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert " bad 0 " in out.stdout
