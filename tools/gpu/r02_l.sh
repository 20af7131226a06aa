# round 2: the tree-split scorer on FULL batches (it was only used for few tiles): wavefronts per tile 1 / 2 / 4 / 8 / 16
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_l}
O=gpurun_out/$TAG
mkdir -p $O
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
for nw in 1 2 4 8 16; do
  env MRK_QS_SPLIT=$nw timeout 600 python bench.py --workload c2 $Q > $O/c2_nw$nw.json 2> $O/c2_nw$nw.log; show "c2 trees split $nw" $O/c2_nw$nw.json
done
for nw in 4 8; do
  env MRK_QS_SPLIT=$nw timeout 600 python bench.py --workload c3 $Q > $O/c3_nw$nw.json 2> $O/c3_nw$nw.log; show "c3 trees split $nw" $O/c3_nw$nw.json
  env MRK_QS_SPLIT=$nw timeout 600 python bench.py --workload c2 --streams 1 $Q > $O/c2s1_nw$nw.json 2> $O/c2s1_nw$nw.log; show "c2 one stream, trees split $nw" $O/c2s1_nw$nw.json
done
env MRK_QS_SPLIT=8 timeout 900 python bench.py --workload c4x --clones 19 --items 2000000 $Q > $O/c4x_nw8.json 2> $O/c4x_nw8.log; show "c4x(2M) trees split 8" $O/c4x_nw8.json
env MRK_QS_SPLIT=8 timeout 600 python bench.py --workload c2 --backend xgboost --trees 500 --depth 4 $Q > $O/c2_xgbd4_nw8.json 2> $O/c2_xgbd4_nw8.log; show "c2 xgb 500 x d4 (f32 bit-vector), split 8" $O/c2_xgbd4_nw8.json
