cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kb in 0 12.25 13; do
MRK_QS_LDS_KB=$kb timeout 900 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lds_kb=$kb V=',d['config']['tile_columns'], round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
