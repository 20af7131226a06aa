cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_score_gpu.py -x -q 2>&1 | tail -2
for rq in 3840 4096 5120 7680 2048; do
timeout 900 python bench.py --requests $rq --steps 20 --warmup 3 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('requests=$rq', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
