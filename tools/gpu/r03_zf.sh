# round 3, pass zf: where a request's time goes in the serving queue (device / host breakdown from mrk_serve_stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_zf
mkdir -p $O
timeout 600 python -m pytest tests/test_rank_one_gpu.py -m gpu -x -q -k "serving_queue" > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -2
timeout 300 python bench.py --steps 2 --warmup 1 --batches-per-step 4 --cpu-sample 0 --latency-requests 400 --e2e-seconds 0 > $O/b.json 2> $O/b.log || tail -3 $O/b.log
python - $O/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(d['latency'], indent=1))
PY
