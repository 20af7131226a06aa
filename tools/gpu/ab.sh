# same-box A/B of library builds under ab/<name>/ (tools/ab_build.sh): VARIANTS="a b c", two rounds, c2 (+ c3 with WL="c2 c3")
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-ab}
mkdir -p $O
for rep in 1 2; do for v in $VARIANTS; do for w in ${WL:-c2}; do
  MRK_LIB=$PWD/ab/$v/libmrk_hip.so timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 --latency-requests ${LAT:-0} --e2e-seconds 0 > $O/${v}_${w}_$rep.json 2> $O/${v}_${w}_$rep.log || tail -3 $O/${v}_${w}_$rep.log
  python - $v $w $rep $O/${v}_${w}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[4]))
    print(sys.argv[1].ljust(14), sys.argv[2], sys.argv[3], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['latency'] and round(d['latency']['p50_ms'], 4))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done; done
