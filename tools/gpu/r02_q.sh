# round 2: same-box A/B of the previous commit's library against the slices build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_q}
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
for rep in 1 2; do
for wl in c2 c3; do
  env MRK_LIB=$PWD/metarank_amd/libmrk_hip_head.so timeout 600 python bench.py --workload $wl $Q > $O/${wl}_head$rep.json 2> $O/${wl}_head$rep.log; show "$wl head" $O/${wl}_head$rep.json
  env MRK_FUSED_SLICES=1 timeout 600 python bench.py --workload $wl $Q > $O/${wl}_new$rep.json 2> $O/${wl}_new$rep.log; show "$wl new, slices off" $O/${wl}_new$rep.json
done
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head
