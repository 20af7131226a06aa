# round 2: counting order for requests of <= 256 candidates, one compare-exchange per lane in the bitonic sorts, four heads per attention workgroup
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_y}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_rank_parity.py tests/test_known_answers.py tests/test_encoder_gpu.py tests/test_serving_loop.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
Q="--steps 5 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); l = d.get('latency')
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'p50', l and round(l['p50_ms'], 3), d.get('encoder') and round(d['encoder']['ms_per_step'], 3))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for w in c2 c3 c4 c5; do
  timeout 600 python bench.py --workload $w $Q > $O/$w.json 2> $O/$w.log; show "$w" $O/$w.json
done
