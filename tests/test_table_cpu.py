"""CPU: the device code of the pre-pass hash tables (csrc/table_device.hpp: insert, lookup, their list forms) compiled for
the HOST with a one-lane stand-in for the wavefront primitives (tests/native/table_test.cpp) - every variant of the lookup
(the default, and the gated experiments MRK_LEAN_GET / MRK_GET_PAIR / both) against a std::map, window widths 2 / 3 / 4 / 8,
tables from 8 entries up, empty to over-full, and six host threads (six one-lane wavefronts) inserting into one table at once;
ASan + UBSan.  The wavefront-level behaviour is the GPU suites' business."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("width", [4, 2, 3, 8])
def test_table_variants_agree_with_a_map(tmp_path, width):
    exe = str(tmp_path / "table_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas",
                           f"-DMRK_PROBE_W={width}", "-I" + os.path.join(REPO, "metarank_amd", "csrc"),
                           os.path.join(REPO, "tests", "native", "table_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"PROBE_W {width}:" in out.stdout and " 0 bad" in out.stdout
