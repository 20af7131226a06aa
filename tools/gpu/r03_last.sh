# round 3, last pass: the whole -m gpu suite, smoke(), the RCCL code path of the bench with a world of one, one default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_last
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for w in c2 c4; do
  MRK_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/dist_$w.json 2> $O/dist_$w.log || tail -5 $O/dist_$w.log
  python - $w $O/dist_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print('world-of-one RCCL path', sys.argv[1], round(d['value']/1e6, 1), 'M items/s', d['config']['parallelism'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log || tail -5 $O/bench_default.log
python - $O/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('default bench line:', round(d['value']/1e6, 1), 'M items/s', 'e2e', round(d['e2e']['value']/1e6, 1), round(d['e2e']['frac_of_value'], 3), 'lat', d['latency']['p50_ms'], 'roofline', round(d['roofline']['frac'], 4), 'valu', {k: round(v['frac'], 2) for k, v in (d['roofline']['valu_issue'] or {}).items()}, 'cpu', round(d['cpu_baseline']['value']))
PY
