# round 3, pass t: register bitonic in the multi-workgroup sort - tests, c4 / c4x, kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r03_t}
mkdir -p $O
timeout 900 python -m pytest tests/test_big_sort_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|error|assert" $O/pytest.log | tail -4
for w in c4 c4x; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_$w.json 2> $O/bench_$w.log || tail -5 $O/bench_$w.log
  python - $w $O/bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python bench.py --workload $w --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_$w.log 2>&1
  grep -E "ss_|Name" $O/stats_$w/*kernel_stats.csv | cut -d, -f1-4 | sed 's/mrk::(anonymous namespace):://' | cut -c1-110
done
find $O -name "*kernel_trace.csv" -size +1M -delete
