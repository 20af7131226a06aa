#!/bin/bash
# c4 (one 100 000-candidate request): parity tests of the big sort + the bench line per chunk size of the LDS chunk sort
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c4
for c in ${CHUNKS:-1024 2048 4096}; do
MRK_SORT_CHUNK=$c timeout 600 python -m pytest tests/test_rank_parity.py -m gpu -x -q -k "c4" 2>&1 | tail -1
MRK_SORT_CHUNK=$c timeout 300 python bench.py --workload c4 --cpu-sample 0 --latency-requests 0 > gpurun_out/c4/bench_$c.json 2> gpurun_out/c4/bench_$c.log || tail -5 gpurun_out/c4/bench_$c.log
python - <<PY
import json
d=json.load(open("gpurun_out/c4/bench_$c.json"))
print("c4 chunk $c", round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
PY
done
