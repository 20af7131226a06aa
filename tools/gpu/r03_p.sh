# round 3, pass p: full batches through ONE kernel (assembly + forest + ordering per request workgroup) vs three launches, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p
mkdir -p $O
timeout 1500 python -m pytest tests/test_rank_parity.py tests/test_serving_loop.py tests/test_known_answers.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rep in 1 2; do for f in 1 0; do for s in 1 2 3; do
  MRK_RANK_FUSED_SCORE=$f timeout 300 python bench.py --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/b_f${f}_s${s}_$rep.json 2> $O/b_f${f}_s${s}_$rep.log || tail -3 $O/b_f${f}_s${s}_$rep.log
  python - $f $s $O/b_f${f}_s${s}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print('fused_score', sys.argv[1], 'streams', sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done; done
