# round 2, sixth GPU pass: the suite with the new paths (norm on device, legacy XGBoost, RCCL world of one, inf on every
# path), config 2 with the specialised matrix kernel, the RCCL leg of bench.py on one GPU, phase clocks of c2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_f}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['config'].get('scorer'), d['config'].get('parallelism'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 600 python bench.py --workload c2 --backend xgboost --trees 100 --depth 6 $Q > $O/c2_xgb_d6.json 2> $O/c2_xgb_d6.log; show "c2 xgb 100x d6" $O/c2_xgb_d6.json
env MRK_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload c2 $Q > $O/c2_dist1.json 2> $O/c2_dist1.log || tail -5 $O/c2_dist1.log; show "c2 through RCCL (world 1)" $O/c2_dist1.json
env MRK_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload c4 $Q > $O/c4_dist1.json 2> $O/c4_dist1.log || tail -5 $O/c4_dist1.log; show "c4 through RCCL (world 1)" $O/c4_dist1.json
timeout 600 python bench.py --workload c4 $Q > $O/c4.json 2> $O/c4.log; show "c4" $O/c4.json
MRK_DEFINES=MRK_PHASE_CLOCKS python -c "from metarank_amd import _native; _native.build(force=True)" > $O/phase_build.log 2>&1
MRK_DEFINES=MRK_PHASE_CLOCKS timeout 600 python tools/phase_clocks.py c2 > $O/phase_c2.txt 2>&1; tail -40 $O/phase_c2.txt
