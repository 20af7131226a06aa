# round 3, pass zd: forest + ordering on a high-priority stream (MRK_SCORE_PRIORITY), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_zd
mkdir -p $O
for rep in 1 2; do for p in 0 1; do for s in 2 3; do
  MRK_SCORE_PRIORITY=$p timeout 300 python bench.py --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/p${p}_s${s}_$rep.json 2> $O/p${p}_s${s}_$rep.log || tail -3 $O/p${p}_s${s}_$rep.log
  python - $p $s $O/p${p}_s${s}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print('priority', sys.argv[1], 'streams', sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch')
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done; done
