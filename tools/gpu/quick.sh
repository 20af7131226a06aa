#!/bin/bash
# quick loop: rank-path GPU parity tests + one c2 bench line (1 and 2 batches in flight)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/quick
( time timeout 900 python -m pytest tests/test_rank_parity.py tests/test_known_answers.py tests/test_write_path.py tests/test_codec.py -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | grep -v "^$\|user\|sys"
run() { tag=$1; shift; timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 256 --latency-requests 100 "$@" > gpurun_out/quick/$tag.json 2> gpurun_out/quick/$tag.log || tail -5 gpurun_out/quick/$tag.log
python - <<PY
import json
d=json.load(open("gpurun_out/quick/$tag.json"))
print("$tag".ljust(10), round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['latency'] and round(d['latency']['p50_ms'],3))
PY
}
run s1 --streams 1
MRK_RANK_JIT=0 run s1_nojit --streams 1
run s2
for w in ${EXTRA_WORKLOADS:-}; do run $w --workload $w; done
