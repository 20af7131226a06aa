# round 3, pass d: cheap knobs on c2 - batches in flight x wavefronts per SIMD of the assembly kernel (co-residency with the scorer)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_d
mkdir -p $O
for w in 4 3; do for s in 2 3 4; do
  MRK_RANK_JIT=1 MRK_JIT_WAVES=$w timeout 300 python bench.py --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/b_w${w}_s$s.json 2> $O/b_w${w}_s$s.log || tail -3 $O/b_w${w}_s$s.log
  python - $w $s $O/b_w${w}_s$s.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print('waves', sys.argv[1], 'streams', sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done
