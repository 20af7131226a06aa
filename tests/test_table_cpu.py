"""CPU: the device code of the pre-pass hash tables (csrc/table_device.hpp: insert, lookup, their list forms) compiled for
the HOST with a one-lane stand-in for the wavefront primitives (tests/native/table_test.cpp) - against a std::map, buckets of
2 / 4 entries, tables from 8 entries up, empty to over-full, and six host threads (six one-lane wavefronts) inserting into one table at once;
ASan + UBSan.  The wavefront-level behaviour is the GPU suites' business."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("width", [4, 2])
def test_table_variants_agree_with_a_map(tmp_path, width):
    exe = str(tmp_path / "table_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas",
                           f"-DMRK_PROBE_W={width}", "-I" + os.path.join(REPO, "metarank_amd", "csrc"),
                           os.path.join(REPO, "tests", "native", "table_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"PROBE_W {width}:" in out.stdout and " 0 bad" in out.stdout


def test_wave_local_median_and_scan_with_64_threads(tmp_path):
    """csrc/wave_device.hpp (the wave-per-section pre-pass's ballot scan and one-value-per-lane LEGACY median) compiled for
    the host: 64 threads = 64 lanes, ballot and LDS ordering point = barriers (tests/native/wave_test.cpp), under
    ThreadSanitizer - a lane reading what another has not yet written is a report, not a flaky number.  780 cases against a
    sorted-array restatement of the percentile, bit for bit."""
    exe = str(tmp_path / "wave_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-ffp-contract=off", "-fsanitize=thread", "-Wno-unknown-pragmas",
                           "-I" + os.path.join(REPO, "metarank_amd", "csrc"), os.path.join(REPO, "tests", "native", "wave_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and not out.stderr.strip(), out.stdout[-2000:] + out.stderr[-3000:]
    assert "780 cases, 0 bad" in out.stdout
