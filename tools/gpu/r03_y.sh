# round 3, pass y: follow-up to pass x (lean profile at 5 wavefronts per SIMD): which part of it pays, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_y
mkdir -p $O
export MRK_RANK_JIT=1 MRK_JIT_SHIPPED=0
run() { tag=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --workload ${WLD:-c2} --streams $st --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds ${E2E:-0} > $O/$tag.json 2> $O/$tag.log || tail -3 $O/$tag.log
  python - $tag $O/$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    e = d.get('e2e')
    print(sys.argv[1].ljust(26), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, e and ('e2e', round(e['value']/1e6,1)))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
L5="MRK_JIT_WAVES=5 MRK_JIT_REGS=0"
run A_default 2 X=1
run C_w5_lean 2 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
run G_w5_lean_3streams 3 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
run H_w5_regs1 2 MRK_THR_STAGE=0 MRK_JIT_WAVES=5 MRK_JIT_REGS=1
run I_w5_lean_staged 2 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
run J_w5_probe8 2 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=8 MRK_PRE_GROUP_BUDGET=24"
run K_w5_budget48 2 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=48"
run L_w4_nostage_regs0 2 MRK_THR_STAGE=0 MRK_JIT_WAVES=4 MRK_JIT_REGS=0
run A_default_again 2 X=1
run C_w5_lean_again 2 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
WLD=c3 run c3_default 2 X=1
WLD=c3 run c3_w5_lean 2 MRK_THR_STAGE=0 $L5 MRK_JIT_DEFINES="MRK_PROBE_W=4 MRK_PRE_GROUP_BUDGET=24"
