# round 2: sanity of the launch-shape refactor (same choices, same bits): scorer + rank parity, c2 / c3 quick lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_zz
mkdir -p $O
timeout 600 python -m pytest tests/test_score_gpu.py tests/test_rank_parity.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -3
Q="--steps 5 --warmup 2 --cpu-sample 0 --e2e-seconds 0 --latency-requests 100"
for w in c2 c3; do
  timeout 300 python bench.py --workload $w $Q > $O/$w.json 2> $O/$w.log
  python - <<PY
import json
d = json.load(open("$O/$w.json"))
print("$w", round(d['value']/1e6, 1), 'M items/s', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, round(d['latency']['p50_ms'], 3))
PY
done
