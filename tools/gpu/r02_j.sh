# round 2: packed encoder batches - encoder tests, c5 bench, encoder micro-benchmark
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_j}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
timeout 900 python bench.py --workload c5 --cpu-sample 0 > $O/bench_c5.json 2> $O/bench_c5.log || tail -5 $O/bench_c5.log
python - <<PY
import json
d = json.load(open("$O/bench_c5.json"))
print("c5", round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['latency'])
print(d['encoder'])
PY
timeout 600 python tools/encoder_bench.py > $O/encoder_bench.log 2>&1; tail -12 $O/encoder_bench.log
