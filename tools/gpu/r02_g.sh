# round 2, seventh GPU pass: multi-entry hash probes - parity of the assembly paths, c2 / c3 / c4x, phase clocks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_g}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests/test_rank_parity.py tests/test_known_answers.py tests/test_serving_loop.py tests/test_write_path.py tests/test_codec.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['config'].get('scorer'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 600 python bench.py --workload c2 $Q > $O/c2.json 2> $O/c2.log; show "c2" $O/c2.json
timeout 600 python bench.py --workload c3 $Q > $O/c3.json 2> $O/c3.log; show "c3" $O/c3.json
timeout 600 python bench.py --workload c4 $Q > $O/c4.json 2> $O/c4.log; show "c4" $O/c4.json
timeout 900 python bench.py --workload c4x --clones 19 --items 2000000 $Q > $O/c4x.json 2> $O/c4x.log; show "c4x(2M of 2M)" $O/c4x.json
MRK_DEFINES=MRK_PHASE_CLOCKS python -c "from metarank_amd import _native; _native.build(force=True)" > $O/phase_build.log 2>&1
MRK_DEFINES=MRK_PHASE_CLOCKS timeout 600 python tools/phase_clocks.py c2 > $O/phase_c2.txt 2>&1; tail -32 $O/phase_c2.txt
