# round 3, pass l: residency of the assembly kernel (LDS request) x batches in flight: does the scorer co-reside?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l
mkdir -p $O
for m in 0 23000 27000 32000 40000; do for s in 2 3; do
  MRK_FUSED_LDS_MIN=$m timeout 300 python bench.py --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/b_m${m}_s$s.json 2> $O/b_m${m}_s$s.log || tail -3 $O/b_m${m}_s$s.log
  python - $m $s $O/b_m${m}_s$s.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print('lds_min', sys.argv[1], 'streams', sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done
