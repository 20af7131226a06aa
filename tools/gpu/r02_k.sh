# round 2: f32 list pool, 16-way tree split, op split of small batches - the suite, c2 (p50) and c5, single-request kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_k}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'lat', d['latency'], 'e2e', d['e2e'] and round(d['e2e']['value']/1e6, 1))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 900 python bench.py --cpu-sample 0 > $O/bench_c2.json 2> $O/bench_c2.log; show c2 $O/bench_c2.json
env MRK_FUSED_SPLIT=1 MRK_QS_SPLIT=8 timeout 900 python bench.py --cpu-sample 0 --e2e-seconds 0 --steps 2 --warmup 1 > $O/bench_c2_nosplit.json 2> $O/bench_c2_nosplit.log; show "c2 (no op split, 8-way trees)" $O/bench_c2_nosplit.json
env MRK_FUSED_SPLIT=2 timeout 900 python bench.py --cpu-sample 0 --e2e-seconds 0 --steps 2 --warmup 1 > $O/bench_c2_split2.json 2> $O/bench_c2_split2.log; show "c2 (op split 2)" $O/bench_c2_split2.json
timeout 900 python bench.py --workload c3 --cpu-sample 0 --e2e-seconds 0 --steps 2 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.log; show c3 $O/bench_c3.json
timeout 900 python bench.py --workload c5 --cpu-sample 0 > $O/bench_c5.json 2> $O/bench_c5.log; show c5 $O/bench_c5.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat -o s -- python bench.py --steps 1 --warmup 1 --batches-per-step 1 --cpu-sample 0 --e2e-seconds 0 --latency-requests 300 > $O/lat.json 2> $O/lat.log
python - <<PY
import csv
for r in csv.DictReader(open("$O/lat/s_kernel_stats.csv")):
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "avg %8.1f us  min %8.1f  max %8.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
