# Round 4, last check: the GPU suite on the committed tree and the default bench line (must quote the committed counters:
# pmc_stale false); smoke().
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_m}
mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.log || tail -5 $O/bench_c2.log
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('value', round(d['value'] / 1e6, 1), 'e2e', round(d['e2e']['value'] / 1e6, 1), round(d['e2e']['frac_of_value'], 3), 'p50', round(d['latency']['p50_ms'], 4))
print('roofline', r['kernel'], round(r['frac'], 4), 'traffic', r['traffic'], 'raw', r['traffic_raw'], 'stale', r['pmc_stale'], 'kernel', r['traffic_kernel'])
print('valu_issue', r['valu_issue'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
timeout 600 python bench.py --workload c4x --cpu-sample 0 > $O/bench_c4x.json 2> $O/bench_c4x.log
python - $O/bench_c4x.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('c4x', round(d['value'] / 1e6, 1), 'roofline', r['kernel'], round(r['frac'], 4), 'stale', r['pmc_stale'], 'gather', r.get('gather'))
PY
