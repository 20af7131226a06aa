# round 3, pass b: the one-launch path of mrk_rank - parity tests, p50 of a 100-item request one-launch vs three-launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_b
mkdir -p $O
timeout 1500 python -m pytest tests/test_rank_one_gpu.py -m gpu -x -q > $O/pytest_one.log 2>&1; tail -15 $O/pytest_one.log
for one in 1 0; do
  MRK_RANK_ONE=$one timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 400 --e2e-seconds 0 > $O/bench_one$one.json 2> $O/bench_one$one.log || tail -5 $O/bench_one$one.log
  python - $one $O/bench_one$one.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print('MRK_RANK_ONE', sys.argv[1], 'latency', d['latency'], 'value', round(d['value']/1e6,1))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
