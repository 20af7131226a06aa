cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for st in 1 2; do
timeout 900 python bench.py --streams $st --steps 40 --warmup 4 --cpu-sample 16 --latency-requests 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams=$st', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['latency'], d['cpu_baseline']['all_cores'])
"
done
for w in c3 c4; do
timeout 900 python bench.py --workload $w --steps 40 --warmup 4 --cpu-sample 4 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
