cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rows in 100 1000 10000 100000; do
for sp in 1 -1; do
MRK_QS_SPLIT=$sp timeout 300 python tools/score_bench.py $rows 24 lgbm 500 2>&1 | tail -1 | sed "s/^/split=$sp /"
done; done
timeout 900 python bench.py --streams 2 --steps 40 --warmup 4 --cpu-sample 16 --latency-requests 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c2', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', d['latency'])
"
timeout 900 python bench.py --workload c4 --steps 40 --warmup 4 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c4', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
