"""TEST INFRASTRUCTURE — compiles the CPU oracle (g++, no GPU code) into oracle/liboracle.so."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["forest_oracle.cpp", "assembly_oracle.cpp"]
LIB = os.path.join(HERE, "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-o", LIB] + srcs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
