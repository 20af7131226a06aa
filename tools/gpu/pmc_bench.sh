cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
ARGS="${BENCH_ARGS:---steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0}"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o s -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p3 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM
run p4 FETCH_SIZE
run p5 WRITE_SIZE
run p6 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py $ARGS > $OUT/stats.log 2>&1
python tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 $OUT/stats > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_bench/summary.json"))
for k,v in d.items():
    if "kernel" in k:
        print(k, {c: round(x.get("mean", x.get("avg_ns",0)),1) for c,x in v.items()})
PY
