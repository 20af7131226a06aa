cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for st in 1 2 3 4; do
timeout 900 python bench.py --streams $st --steps 40 --warmup 4 --cpu-sample 16 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams=$st', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
timeout 900 python bench.py --workload c3 --streams 2 --steps 40 --warmup 4 --cpu-sample 8 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c3 streams=2', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms')
"
