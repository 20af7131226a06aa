# round 3, pass x: lean profiles of the c2 assembly kernel for more resident workgroups (no LDS staging of thresholds, fewer registers), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_x
mkdir -p $O
export MRK_RANK_JIT=1 MRK_JIT_SHIPPED=0
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --workload ${WLD:-c2} --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/$tag.json 2> $O/$tag.log || tail -3 $O/$tag.log
  python - $tag $O/$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1].ljust(22), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for rep in 1 2; do
run A_default_$rep X=1
run B_nostage_w4_$rep MRK_THR_STAGE=0
run C_w5_$rep MRK_THR_STAGE=0 MRK_JIT_WAVES=5 MRK_JIT_REGS=0 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
run D_w6_$rep MRK_THR_STAGE=0 MRK_JIT_WAVES=6 MRK_JIT_REGS=0 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=16"
run E_w8_$rep MRK_THR_STAGE=0 MRK_JIT_WAVES=8 MRK_JIT_REGS=0 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=8"
done
