# round 2, last call: the whole -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_z}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.log; tail -c 600 $O/bench_c2.json | head -c 600; echo
python - <<PY
import json
d = json.load(open("$O/bench_c2.json"))
print(round(d['value']/1e6,1), 'M items/s', d['ms_per_step'], 'e2e', round(d['e2e']['value']/1e6,1), d['latency'], d['roofline']['frac'], d['roofline']['traffic_kernel'], d['cpu_baseline']['value'])
PY
