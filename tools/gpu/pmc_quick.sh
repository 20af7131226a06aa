#!/bin/bash
# two PMC passes (instruction mix, wait/active cycles) of a short c2 bench run; prints the assembly kernel's counters per wavefront
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
ARGS="--streams 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o s -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT
P3=$OUT/p3; ls $OUT/p3/*/*counter_collection.csv >/dev/null 2>&1 || ls $OUT/p3/*counter_collection.csv >/dev/null 2>&1 || P3=
python tools/pmc_summary.py $OUT/p1 $OUT/p2 $P3 > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_quick/summary.json"))
for k,v in d.items():
    if "rank" in k or "qs_score" in k:
        w = v.get("SQ_WAVES", {}).get("mean", 1) or 1
        print(k[:60], "waves", w, {c: round(x.get("mean", 0) / w, 1) for c, x in v.items() if c != "SQ_WAVES"})
PY
