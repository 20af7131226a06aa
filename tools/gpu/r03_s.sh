# round 3, pass s: sort kernel with LDS sized for the batch - serving loop (e2e) and bench line; quick parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r03_s}
mkdir -p $O
timeout 900 python -m pytest tests/test_rank_parity.py tests/test_serving_loop.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -2
for nb in 3 4; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 1.5 --e2e-batches $nb > $O/b_n$nb.json 2> $O/b_n$nb.log || tail -3 $O/b_n$nb.log
  python - $nb $O/b_n$nb.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    e = d['e2e']
    print('in flight', sys.argv[1], 'value', round(d['value']/1e6, 1), {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'e2e', round(e['value']/1e6, 1), 'frac', round(e['frac_of_value'], 3), {k: round(v, 3) for k, v in e['host_ms_per_batch'].items()})
except Exception as ex:
    print(sys.argv[1], 'FAILED', ex)
PY
done
