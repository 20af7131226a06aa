cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_i; rm -rf $OUT; mkdir -p $OUT
i=0
for drop in "divers_year,divers_popularity" "divers_genres,divers_actors,divers_tags" "divers_year" "profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity,ctr,ctr_tag,ctr_genre"; do
i=$((i+1))
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p$i -o s -- python bench.py --drop-features "$drop" --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 > $OUT/p$i.log 2>&1
python tools/pmc_summary.py $OUT/p$i > $OUT/s$i.json
python - <<PY
import json
d=json.load(open("$OUT/s$i.json"))
for k,v in d.items():
    if "fused_cells" in k: print("drop=[$drop]", {c: round(x.get("mean", 0)/7680,0) for c,x in v.items() if 'INSTS' in c or 'CYCLES' in c})
PY
done
