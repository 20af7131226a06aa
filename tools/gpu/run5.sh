cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
MRK_RANK_FUSED=$1 MRK_RANK_CELLS=$2 timeout 600 python bench.py --steps 20 --warmup 3 --cpu-sample 64 --latency-requests 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fused=$1 cells=$2', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['latency'])
"
done
