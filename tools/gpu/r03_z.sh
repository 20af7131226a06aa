# round 3, pass z: the 96-register profile of the full-batch assembly kernel: staging x fetch-ahead budget, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_z
mkdir -p $O
export MRK_RANK_JIT=1 MRK_JIT_SHIPPED=0
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/$tag.json 2> $O/$tag.log || tail -3 $O/$tag.log
  python - $tag $O/$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1].ljust(26), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
L5="MRK_JIT_WAVES=5 MRK_JIT_REGS=0"
run A_default X=1
for b in 24 48 72; do
run stage1_b$b $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=$b"
run stage0_b$b MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=$b"
done
run w5_only MRK_JIT_WAVES=5
run w5_probe4 MRK_JIT_WAVES=5 MRK_JIT_DEFINES="MRK_PROBE_W=4"
run w4_probe4 MRK_JIT_DEFINES="MRK_PROBE_W=4"
run A_default_again X=1
