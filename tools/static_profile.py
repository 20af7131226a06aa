#!/usr/bin/env python
"""Instructions of a run-time specialised kernel by source function - without a device.  The translation unit jit.cpp hands
to hiprtc (mrk_config_specialize, what = 0) is compiled offline with the same flags plus -gline-tables-only, the gfx950
object disassembled with source lines (llvm-objdump -d -l) and every instruction attributed to the function its line lies
in (the innermost inlined function).  `python tools/static_profile.py [c2|c3] [kernel 1..7] [MRK_JIT_DEFINES...]`
(profiles/r03_static_profile_c2.txt is its output for the stock program, annotated)."""
import collections
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3:
    os.environ["MRK_JIT_DEFINES"] = " ".join(sys.argv[3:])
from metarank_amd import _native
from workloads import ranklens

cfg = ranklens.c3_config() if wl == "c3" else ranklens.ranklens_config()
lib = _native.lib()
js = json.dumps(cfg).encode()
need = C.c_size_t(0)
what = 0 | (kernel << 8)
if os.environ.get("MRK_STATIC_MODEL"):   # keyed by the view signature of a bench-like forest too (tools/jit_inspect.py --model)
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from jit_inspect import bench_like_model
    mb = bench_like_model(cfg, wl)
    lib.mrk_config_specialize_for_model(js, len(js), b"xgboost", 0, mb, len(mb), what, None, 0, C.byref(need))
    buf = (C.c_uint8 * need.value)()
    assert lib.mrk_config_specialize_for_model(js, len(js), b"xgboost", 0, mb, len(mb), what, buf, need.value, C.byref(need)) == 0
else:
    lib.mrk_config_specialize(js, len(js), b"xgboost", 1, what, None, 0, C.byref(need))
    buf = (C.c_uint8 * need.value)()
    assert lib.mrk_config_specialize(js, len(js), b"xgboost", 1, what, buf, need.value, C.byref(need)) == 0
tmp = tempfile.mkdtemp(prefix="mrk_static_")
tu = os.path.join(tmp, "tu.hip")
with open(tu, "wb") as f:
    f.write(b"#include <hip/hip_runtime.h>\n" + bytes(buf)[:need.value])
LLVM = "/opt/rocm/lib/llvm/bin/"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", "--cuda-device-only",
                       "-w", "-c", tu, "-o", os.path.join(tmp, "tu.o")])
subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + os.path.join(tmp, "tu.o"),
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(tmp, "dev.o")])
asm = subprocess.run([LLVM + "llvm-objdump", "-d", "-l", "--no-show-raw-insn", os.path.join(tmp, "dev.o")], capture_output=True, text=True).stdout
src = open(tu).read().split("\n")
funcs = []
for i, l in enumerate(src, 1):
    m = re.match(r"^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?__device__\s+(?:__forceinline__\s+)?(?:constexpr\s+)?[\w:<>\*&\s]+?\b(\w+)\s*\(", l)
    if m and not l.strip().startswith("//"):
        funcs.append((i, m.group(1)))
    if l.startswith('extern "C" __global__'):
        funcs.append((i, "<kernel>"))


def func_of(line):
    name = "?"
    for ln, n in funcs:
        if ln > line:
            break
        name = n
    return name


cur = None
per_line, per_func, kinds = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
for l in asm.split("\n"):
    if l.startswith("; " + tu + ":"):
        cur = int(l.strip().split(":")[-1])
        continue
    if l.startswith(";") or not l.startswith("\t") or cur is None:
        continue
    op = l.split()[0]
    f = func_of(cur)
    per_line[cur] += 1
    per_func[f] += 1
    kinds[f]["S" if op.startswith("s_") else "V" if op.startswith("v_") else "M"] += 1
tot = sum(per_func.values())
print(f"{wl} kernel {kernel} defines [{os.environ.get('MRK_JIT_DEFINES', '')}]: {tot} instructions")
for f, c in per_func.most_common(20):
    print(f"  {f:30s} {c:6d} {100 * c / tot:5.1f} %   VALU {kinds[f]['V']:5d}  SALU {kinds[f]['S']:5d}  memory {kinds[f]['M']:4d}")
print("lines:")
for ln, c in per_line.most_common(15):
    print(f"  {ln:5d} {c:5d}  {func_of(ln):22s} {src[ln - 1].strip()[:100]}")
if os.environ.get("MRK_STATIC_KEEP"):   # the annotated disassembly, for a closer look
    open(os.environ["MRK_STATIC_KEEP"], "w").write(asm)
