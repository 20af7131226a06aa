# round 3, pass za: probe window of the pre-pass hash tables (new window loop) on c2 and c3, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_za
mkdir -p $O
export MRK_RANK_JIT=1 MRK_JIT_SHIPPED=0
run() { tag=$1; w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 --latency-requests ${LAT:-0} --e2e-seconds 0 > $O/${tag}_$w.json 2> $O/${tag}_$w.log || tail -3 $O/${tag}_$w.log
  python - $tag $w $O/${tag}_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print(sys.argv[1].ljust(22), sys.argv[2], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['latency'] and round(d['latency']['p50_ms'], 4))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for w in c2 c3; do
run default $w X=1
for p in 2 3 4 6; do run probe$p $w MRK_JIT_DEFINES="MRK_PROBE_W=$p"; done
run probe4_w5_regs0 $w MRK_JIT_WAVES=5 MRK_JIT_REGS=0 MRK_JIT_DEFINES="MRK_PROBE_W=4"
run probe4_regs0 $w MRK_JIT_REGS=0 MRK_JIT_DEFINES="MRK_PROBE_W=4"
done
LAT=200 run default_lat c2 X=1
LAT=200 run probe4_lat c2 MRK_JIT_DEFINES="MRK_PROBE_W=4"
