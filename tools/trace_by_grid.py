#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> mean duration per (kernel, grid size): separates the big-batch launches of a kernel
from its single-request launches, which `--stats` averages together.  python tools/trace_by_grid.py <kernel_trace.csv> [min_us]"""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^(void )?(mrk::)?(\(anonymous namespace\)::)?", "", name)
    m = re.match(r"_ZN3mrk12_GLOBAL__N_1\d+([a-z_]+)(I.*?E)EvP", name)
    if m:
        name = m.group(1) + "<" + ",".join(re.findall(r"Li(\d+)E", m.group(2))) + ">"
    rows[(name[:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':60s} {'workgroups':>16s} {'calls':>6s} {'mean us':>9s} {'total ms':>9s} {'%':>5s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / len(v) < min_us:
        continue
    print(f"{k[0]:60s} {str(k[1:]):>16s} {len(v):6d} {sum(v) / len(v):9.1f} {sum(v) / 1e3:9.3f} {100 * sum(v) / tot:5.1f}")
