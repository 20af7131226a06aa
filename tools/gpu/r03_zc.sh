# round 3, pass zc: fetch-ahead widths of the token ops (MRK_JIT_DEFINES), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_zc
mkdir -p $O
export MRK_RANK_JIT=1 MRK_JIT_SHIPPED=0
run() { tag=$1; w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/${tag}_$w.json 2> $O/${tag}_$w.log || tail -3 $O/${tag}_$w.log
  python - $tag $w $O/${tag}_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print(sys.argv[1].ljust(22), sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run default c2 X=1
run tok4 c2 MRK_JIT_DEFINES="MRK_TOK_BATCH=4"
run tok6 c2 MRK_JIT_DEFINES="MRK_TOK_BATCH=6"
run iw2 c2 MRK_JIT_DEFINES="MRK_IW_TOK=2"
run iw6 c2 MRK_JIT_DEFINES="MRK_IW_TOK=6"
run b48 c2 MRK_JIT_DEFINES="MRK_PRE_GROUP_BUDGET=48"
run b96 c2 MRK_JIT_DEFINES="MRK_PRE_GROUP_BUDGET=96"
run tok4_iw6 c2 MRK_JIT_DEFINES="MRK_TOK_BATCH=4 MRK_IW_TOK=6"
run default_again c2 X=1
