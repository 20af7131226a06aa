# round 2: stand-alone pre-pass kernel with its tables in LDS (c4): parity tests + A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_x}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_rank_parity.py tests/test_known_answers.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for l in 1 0 1 0; do
  env MRK_PREPASS_LDS=$l timeout 600 python bench.py --workload c4 $Q > $O/c4_l$l.json 2> $O/c4_l$l.log; show "c4, pre-pass tables in LDS=$l" $O/c4_l$l.json
done
env MRK_RANK_FUSED=0 timeout 600 python bench.py --workload c2 $Q > $O/c2_unfused.json 2> $O/c2_unfused.log; show "c2 unfused, LDS pre-pass" $O/c2_unfused.json
env MRK_RANK_FUSED=0 MRK_PREPASS_LDS=0 timeout 600 python bench.py --workload c2 $Q > $O/c2_unfused_l0.json 2> $O/c2_unfused_l0.log; show "c2 unfused, HBM pre-pass" $O/c2_unfused_l0.json
