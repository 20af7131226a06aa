# round 2, eighth GPU pass: what a single mrk_rank spends on the device (kernel trace of the latency leg); occupancy of
# the item-parallel specialised kernel on the out-of-cache gather
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_h}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python bench.py --steps 1 --warmup 1 --batches-per-step 1 --cpu-sample 0 --e2e-seconds 0 --latency-requests 300 > $O/lat_warm.json 2> $O/lat_warm.log   # warms the JIT cache
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat -o s -- python bench.py --steps 1 --warmup 1 --batches-per-step 1 --cpu-sample 0 --e2e-seconds 0 --latency-requests 300 > $O/lat.json 2> $O/lat.log
python - <<PY
import csv, json
d = json.load(open("$O/lat_warm.json")); print("latency (no profiler)", d["latency"])
for r in csv.DictReader(open("$O/lat/s_kernel_stats.csv")):
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "avg %8.1f us  min %8.1f  max %8.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
for w in 3 5 6; do
  env MRK_JIT_WAVES=$w timeout 900 python bench.py --workload c4x --clones 19 --items 2000000 $Q > $O/c4x_w$w.json 2> $O/c4x_w$w.log
  python - <<PY
import json
d = json.load(open("$O/c4x_w$w.json")); print("c4x(2M) waves $w", {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
PY
done
