cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r01_i}
mkdir -p gpurun_out/$TAG
for w in c2 c3 c4 c5; do
timeout 900 python bench.py --workload $w > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.log || tail -5 gpurun_out/$TAG/bench_$w.log
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_$w.json"))
print("$w", round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['latency'], d['roofline']['kernel'], round(d['roofline']['frac'],4), d['roofline']['traffic'], d['cpu_baseline'] and (round(d['cpu_baseline']['value']), d['cpu_baseline']['all_cores']))
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o s -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 > gpurun_out/$TAG/stats.log 2>&1
cut -c1-160 gpurun_out/$TAG/stats/s_kernel_stats.csv | head -8
BENCH_ARGS="--streams 1 --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0" bash tools/gpu/pmc_bench.sh | cut -c1-1200
cp gpurun_out/pmc_bench/summary.json gpurun_out/$TAG/pmc_summary.json
