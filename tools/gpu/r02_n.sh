# round 2: the suite after the scorer-residency change; where the serving loop's host time goes (1 / 2 / 3 host threads)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_n}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
for thr in 1 2 3; do
  timeout 600 python bench.py --cpu-sample 0 --latency-requests 0 --steps 3 --warmup 1 --e2e-threads $thr > $O/e2e_t$thr.json 2> $O/e2e_t$thr.log
  python - <<PY
import json
d = json.load(open("$O/e2e_t$thr.json")); e = d["e2e"]
print("c2 threads $thr: value", round(d["value"]/1e6,1), "e2e", round(e["value"]/1e6,1), "ms/batch", round(e["ms_per_batch"],3), {k: round(v,3) for k,v in e["host_ms_per_batch"].items()})
PY
done
timeout 600 python bench.py --backend xgboost --trees 100 --depth 6 --cpu-sample 0 --latency-requests 0 --steps 3 --warmup 1 --e2e-threads 1 > $O/e2e_xgb_t1.json 2> $O/e2e_xgb_t1.log
python - <<PY
import json
d = json.load(open("$O/e2e_xgb_t1.json")); e = d["e2e"]
print("config 2 threads 1: value", round(d["value"]/1e6,1), "e2e", round(e["value"]/1e6,1), "ms/batch", round(e["ms_per_batch"],3), {k: round(v,3) for k,v in e["host_ms_per_batch"].items()})
PY
