"""Compare the machine code of two sets of specialised-kernel code objects kernel by kernel (llvm-objdump -d, addresses
stripped): `python tools/co_diff.py DIR_A DIR_B` - which kernels exist on both sides and whether their instruction streams
are identical.  Used to show that a refactoring or an `#ifdef`-gated experiment leaves the default kernels untouched.
With several code objects of one kernel in a directory (older builds), the NEWEST file wins."""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(d):
    out = {}
    for f in sorted((f for f in os.listdir(d) if f.endswith(".co")), key=lambda f: os.path.getmtime(os.path.join(d, f))):
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(d, f)], capture_output=True, text=True).stdout
        name, body = None, []
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                if name:
                    out[name] = (f, body)
                name, body = m.group(1), []
            elif name and line.strip():
                body.append(re.sub(r"//.*$", "", line).strip())
        if name:
            out[name] = (f, body)
    return out


if __name__ == "__main__":
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    rc = 0
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b:
            print(f"{k}: only in {'A' if k in a else 'B'}")
            continue
        same = a[k][1] == b[k][1]
        rc |= 0 if same else 1
        print(f"{k}: {'identical' if same else 'DIFFERENT'} ({len(a[k][1])} vs {len(b[k][1])} instructions; {a[k][0]} / {b[k][0]})")
    sys.exit(rc)
