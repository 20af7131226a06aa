cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cat in 100000 10000 2000; do
timeout 900 python bench.py --catalogue $cat --sessions $((cat/10)) --streams 1 --steps 20 --warmup 3 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('catalogue=$cat', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
