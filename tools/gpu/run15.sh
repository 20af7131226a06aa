cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_rank_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c4', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
