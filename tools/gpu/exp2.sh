#!/bin/bash
# experiments: one bench line per argument string (extra bench.py arguments)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/exp
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 64 --latency-requests 0 $a > gpurun_out/exp/$i.json 2> gpurun_out/exp/$i.log || tail -5 gpurun_out/exp/$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/exp/$i.json"))
print("$a".ljust(34), round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
PY
done
