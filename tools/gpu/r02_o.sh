# round 2: 512-row tiles of the tree-walk scorer (config 2), batches in flight of the serving loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_o}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_score_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -3
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); e = d.get("e2e")
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()},
          e and ("e2e", round(e["value"]/1e6, 1), {k: round(v, 3) for k, v in e["host_ms_per_batch"].items()}))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for t in 512 256; do
  env MRK_WALK_TILE=$t timeout 600 python bench.py --backend xgboost --trees 100 --depth 6 $Q > $O/xgb6_t$t.json 2> $O/xgb6_t$t.log; show "config 2, walk tile $t" $O/xgb6_t$t.json
  env MRK_WALK_TILE=$t timeout 600 python bench.py --backend xgboost --trees 100 --depth 8 $Q > $O/xgb8_t$t.json 2> $O/xgb8_t$t.log; show "xgb depth 8, walk tile $t" $O/xgb8_t$t.json
done
for nb in 2 4 6; do
  timeout 600 python bench.py --cpu-sample 0 --latency-requests 0 --steps 3 --warmup 1 --e2e-batches $nb > $O/e2e_nb$nb.json 2> $O/e2e_nb$nb.log; show "c2 e2e, $nb batches in flight" $O/e2e_nb$nb.json
done
