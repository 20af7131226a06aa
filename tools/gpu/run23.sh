cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_j; rm -rf $OUT; mkdir -p $OUT
MRK_RANK_FUSED=0 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p1 -o s -- python bench.py --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 > $OUT/p1.log 2>&1
python tools/pmc_summary.py $OUT/p1 > $OUT/s1.json
python - <<PY
import json
d=json.load(open("$OUT/s1.json"))
for k,v in d.items():
    if "prepass" in k or "assemble" in k:
        w=v['SQ_WAVES']['mean']
        print(k, 'waves', w, {c: round(x.get("mean", 0)/w,0) for c,x in v.items() if 'INSTS' in c or 'CYCLES' in c})
PY
