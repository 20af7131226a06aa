cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 3 --cpu-sample 16 --latency-requests 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['latency'])
"
