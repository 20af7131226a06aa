#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c5
timeout 1200 python bench.py --workload c5 > gpurun_out/c5/bench_c5.json 2> gpurun_out/c5/bench_c5.log || tail -20 gpurun_out/c5/bench_c5.log
python - <<PY
import json
d=json.load(open("gpurun_out/c5/bench_c5.json"))
print(round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', d['latency'], d['encoder'], {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['cpu_baseline'] and round(d['cpu_baseline']['value']))
PY
