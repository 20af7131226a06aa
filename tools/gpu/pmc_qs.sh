cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export MRK_QS_R=${MRK_QS_R:-2} MRK_QS_PIPE=${MRK_QS_PIPE:-0}
OUT=gpurun_out/pmc_qs
mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o s -- python tools/score_bench.py 384000 24 lgbm 500 > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p3 GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_qs/summary.json"))
for k,v in d.items():
    if "qs_" in k or "score" in k:
        print(k, {c: round(x.get("mean", x.get("avg_ns",0)),1) for c,x in v.items()})
PY
