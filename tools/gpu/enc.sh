#!/bin/bash
# encoder leg: GPU tests, micro-benchmark, kernel stats
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/enc_prof
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/enc_test.log
cat gpurun_out/enc_test.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_prof -o enc -- python tools/encoder_bench.py > gpurun_out/enc_bench.log 2>&1
grep -v "^[WE]2026" gpurun_out/enc_bench.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/enc_prof/enc_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), r["Percentage"].rjust(7), "avg", r["AverageNs"].split(".")[0].rjust(8), "min", r["MinNs"].rjust(7), "max", r["MaxNs"].rjust(8))
PY
rm -f gpurun_out/enc_prof/enc_kernel_trace.csv gpurun_out/enc_prof/*.db
