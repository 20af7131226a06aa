# round 2, first GPU pass: the whole -m gpu suite, then the default bench line (c2) with its e2e leg
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_a}
mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; tail -15 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench_c2.json 2> gpurun_out/$TAG/bench_c2.log || tail -20 gpurun_out/$TAG/bench_c2.log
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_c2.json"))
print("c2", round(d['value']/1e6,1),'M items/s', round(d['ms_per_device_batch'],3),'ms/batch', {k:round(v['avg_ms']*v['launches_per_batch'],3) for k,v in d['kernels'].items()}, d['latency'], d['roofline']['kernel'], round(d['roofline']['frac'],4))
print("e2e", d['e2e'] and {k: d['e2e'][k] for k in ('value','ms_per_batch','frac_of_value','device_batches')})
print("cpu", d['cpu_baseline'] and (round(d['cpu_baseline']['value']), d['cpu_baseline']['all_cores']))
PY
