# round 3, pass f: load factor of the pre-pass hash tables (probe chains of misses vs LDS per workgroup) on c2 / c3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_f
mkdir -p $O
for w in c2 c3; do for p in 75 60 50 40 30; do
  MRK_TABLE_LOAD_PCT=$p timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/b_${w}_p$p.json 2> $O/b_${w}_p$p.log || tail -3 $O/b_${w}_p$p.log
  python - $w $p $O/b_${w}_p$p.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print(sys.argv[1], 'load pct', sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done
