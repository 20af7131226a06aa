# round 3, pass h: 8-wide probe windows of the pre-pass hash tables - parity, then c2 / c3 / single-request latency
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_h
mkdir -p $O
timeout 1800 python -m pytest tests/test_rank_parity.py tests/test_known_answers.py tests/test_rank_one_gpu.py tests/test_write_path.py tests/test_serving_loop.py -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for w in c2 c3; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 300 --e2e-seconds 0 > $O/b_$w.json 2> $O/b_$w.log || tail -3 $O/b_$w.log
  python - $w $O/b_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'lat', d['latency'] and (round(d['latency']['p50_ms'],4), d['latency'].get('serve_queue')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
