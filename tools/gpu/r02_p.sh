# round 2: slices per request of the fused kernel (c3: few large requests)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_p}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_rank_parity.py tests/test_serving_loop.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); e = d.get("e2e")
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()},
          e and ("e2e", round(e["value"]/1e6, 1), {k: round(v, 3) for k, v in e["host_ms_per_batch"].items()}))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for s in 0 1 2 3 4; do
  env MRK_FUSED_SLICES=$s timeout 600 python bench.py --workload c3 $Q > $O/c3_s$s.json 2> $O/c3_s$s.log; show "c3, slices $s" $O/c3_s$s.json
done
timeout 600 python bench.py --workload c2 $Q > $O/c2.json 2> $O/c2.log; show "c2" $O/c2.json
env MRK_FUSED_SLICES=2 MRK_FUSED_THREADS=64 timeout 600 python bench.py --workload c2 $Q > $O/c2_s2.json 2> $O/c2_s2.log; show "c2, 64 lanes x 2 slices" $O/c2_s2.json
