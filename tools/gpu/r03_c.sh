# round 3, pass c: the serving queue (persistent workgroups) - parity tests under a watchdog, p50 through the queue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_c
mkdir -p $O
timeout 600 python -m pytest tests/test_rank_one_gpu.py -m gpu -x -q -k "serving_queue" > $O/pytest_serve.log 2>&1; tail -25 $O/pytest_serve.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 400 --e2e-seconds 0 > $O/bench_serve.json 2> $O/bench_serve.log || tail -5 $O/bench_serve.log
python - $O/bench_serve.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('latency', d['latency'], 'value', round(d['value']/1e6,1))
except Exception as e:
    print('FAILED', e)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lat -o s -- python bench.py --steps 1 --warmup 1 --batches-per-step 2 --cpu-sample 0 --latency-requests 200 --e2e-seconds 0 > $O/stats_lat.log 2>&1
grep -E "rank_one|rank_serve|Name" $O/stats_lat/*kernel_stats.csv | cut -c1-160
