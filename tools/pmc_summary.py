"""Summarise rocprofv3 CSV output: per-kernel mean counter values / durations.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write ... > profiles/rNN_pmc_summary.json
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


out = {}
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    for f in glob.glob(f"{d}/*_kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            out.setdefault(short(row["Name"]), {})["duration"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                                                  "pct": float(row["Percentage"])}
# which code the counters belong to: the `provenance` object of bench.py's JSON line in each pass's captured output
# (<dir>.log next to <dir>); bench.py quotes a committed summary only when it is running the same build and kernels
provs = []
for d in sys.argv[1:]:
    log = d.rstrip("/") + ".log"
    if os.path.exists(log):
        for ln in open(log, errors="replace"):
            if ln.startswith('{"metric"'):
                try:
                    provs.append(json.loads(ln).get("provenance"))
                except ValueError:
                    pass
provs = [p for p in provs if p]
if provs:
    out["_provenance"] = dict(provs[0], consistent=all(p == provs[0] for p in provs), passes=len(provs))
json.dump(out, sys.stdout, indent=1)
