# r06_w: a forest with ~250 thresholds per column (bench.py --quantiles 254: a LightGBM model trained with max_bin 255) beside the
# benchmark's (49 candidates per column), same box: which sink each kernel runs is decided by the size of the compact tables
O=gpurun_out/r06_w; mkdir -p $O
for wl in c2 c3 c4x; do
for qn in 49 254; do
  timeout 600 python bench.py --workload $wl --quantiles $qn --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 200 --latency-sweep 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl quantiles=$qn', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()}, 'p50', (d.get('latency') or {}).get('p50_ms'), 'tile columns', d['config'].get('tile_columns'))"
done
done | tee $O/ab.txt
