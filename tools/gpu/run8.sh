cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in c2 c3; do
timeout 900 python bench.py --workload $w --steps 20 --warmup 3 --cpu-sample 16 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
done
OUT=gpurun_out/pmc_f; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o s -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run p2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run p3 TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
python tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_f/summary.json"))
for k,v in d.items():
    if "fused" in k: print(k, {c: round(x.get("mean", 0),1) for c,x in v.items()})
PY
tail -3 $OUT/p3.log
