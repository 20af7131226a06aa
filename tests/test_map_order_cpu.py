"""CPU: an `interacted_with` feature over more than 4 fields orders its columns like the reference's Scala immutable Map
iterates (InteractedWithFeature.scala:56-65,152-162; Scala 2.13.16, build.sbt:6): Map1..Map4 keep insertion order, beyond
that `toMap` gives a HashMap whose order is a function of the keys' hashes.  Two restatements - the library's
(csrc/features.cpp scala_map_key_order: groups by hash bits) and the oracle's (oracle/assembly.py: inserts into a prefix
tree node by node and walks it) - against each other on random key sets and against the answers a Scala 2.13 REPL is known
to print: Map("a"->1,…,"e"->5) iterates e, a, b, c, d.  No JVM here: beyond those known answers the order is unpinned, and a
host that sees another order passes it as "field_order"."""
import os
import random
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.assembly import _java_hash, _scala_improve, scala_map_key_order


import pytest


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return build(tmp_path_factory.mktemp("map_order"))


def build(tmp_path):
    from metarank_amd import _native

    _native.build()
    exe = str(tmp_path / "map_order_test")
    lib_dir = os.path.dirname(_native.LIB_PATH)
    csrc = os.path.join(REPO, "metarank_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", os.path.join(REPO, "tests", "native", "map_order_test.cpp"), os.path.join(csrc, "store.cpp"),
                           os.path.join(csrc, "features.cpp"), "-I" + csrc, "-I" + os.path.join(REPO, "include"), "-L" + lib_dir, "-lmrk_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def native(exe, keys, config=False):
    out = subprocess.run([exe] + (["--config"] if config else []) + keys, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout.split("\n")[:-1]


def test_known_answers_of_a_scala_repl():
    assert scala_map_key_order(list("abcd")) == list("abcd")      # Map4: insertion order
    assert scala_map_key_order(list("dcba")) == list("dcba")
    assert scala_map_key_order(list("abcde")) == list("eabcd")    # HashMap(e -> 5, a -> 1, b -> 2, c -> 3, d -> 4)
    assert scala_map_key_order(list("edcba")) == list("eabcd")    # canonical: not a function of the insertion order
    # the same tree over Int keys (## of an Int is the Int): (1 to 5).map(i => i -> i).toMap prints HashMap(5 -> 5, 1 -> 1, 2 -> 2, 3 -> 3, 4 -> 4)
    assert sorted(range(1, 6), key=lambda i: _scala_improve(i) & 31) == [5, 1, 2, 3, 4]
    assert _java_hash("Aa") == _java_hash("BB") == 2112           # String.hashCode's classic collision
    assert _java_hash("metarank") == 0xFFFFFFFF & sum(ord(c) * 31 ** (7 - i) for i, c in enumerate("metarank"))


def test_library_and_oracle_agree(exe):
    assert native(exe, list("abcde")) == list("eabcd")
    rng = random.Random(7)
    alphabet = "abcdefghijklmnopqrstuvwxyz_0123456789"
    for trial in range(60):
        n = rng.randint(1, 40)
        keys = list({"".join(rng.choice(alphabet) for _ in range(rng.randint(1, 12))) for _ in range(n)})
        rng.shuffle(keys)
        assert native(exe, keys) == scala_map_key_order(keys), keys
    # equal hashes ("Aa" / "BB", and every concatenation of such pairs): a collision node keeps insertion order
    for keys in (["x1", "BB", "x2", "Aa", "x3", "x4"], ["AaAa", "BBBB", "AaBB", "BBAa", "q", "r"], ["BBBB", "q", "AaBB", "r", "BBAa", "AaAa"]):
        got = native(exe, keys)
        assert got == scala_map_key_order(keys), keys
        same = [k for k in got if _java_hash(k) == _java_hash(keys[1])]
        assert same == [k for k in keys if _java_hash(k) == _java_hash(keys[1])]
    # keys outside ASCII and outside the BMP hash by UTF-16 code unit
    keys = ["жанр", "演员", "tag", "🎬", "año", "x"]
    assert native(exe, keys) == scala_map_key_order(keys)


def test_a_config_with_six_fields_loads_in_that_order(exe):
    """round 2 answered MRK_ERR_UNSUPPORTED without an explicit field_order"""
    fields = ["genres", "actors", "tags", "director", "writer", "year"]
    assert native(exe, fields, config=True) == scala_map_key_order(fields) != fields
    assert native(exe, fields[:4], config=True) == fields[:4]
