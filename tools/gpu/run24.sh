cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pct in 75 50 33 25; do
MRK_TABLE_LOAD_PCT=$pct timeout 900 python bench.py --streams 1 --steps 30 --warmup 4 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('load=$pct%', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
