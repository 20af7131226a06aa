# round 2: slices per request x batches in flight (c3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_r}
O=gpurun_out/$TAG
mkdir -p $O
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); e = d.get("e2e")
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for st in 1 2; do
for s in 1 2 3 4; do
  env MRK_FUSED_SLICES=$s timeout 600 python bench.py --workload c3 --streams $st $Q > $O/c3_st${st}_s$s.json 2> $O/c3_st${st}_s$s.log; show "c3, streams $st, slices $s" $O/c3_st${st}_s$s.json
done
done
env MRK_FUSED_SLICES=2 timeout 600 python bench.py --workload c3 --requests 96 --streams 1 $Q > $O/c3_96_s2.json 2> $O/c3_96_s2.log; show "c3 96 req, slices 2" $O/c3_96_s2.json
env MRK_FUSED_SLICES=4 timeout 600 python bench.py --workload c3 --requests 96 --streams 1 $Q > $O/c3_96_s4.json 2> $O/c3_96_s4.log; show "c3 96 req, slices 4" $O/c3_96_s4.json
env MRK_FUSED_SLICES=1 timeout 600 python bench.py --workload c3 --requests 96 --streams 1 $Q > $O/c3_96_s1.json 2> $O/c3_96_s1.log; show "c3 96 req, slices 1" $O/c3_96_s1.json
