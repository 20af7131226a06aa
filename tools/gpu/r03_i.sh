# round 3, pass i: instruction / wait counters of the c2 assembly kernel by feature family (bench.py --drop-features)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_i
rm -rf $O; mkdir -p $O
ARGS="--streams 1 --steps 2 --warmup 1 --batches-per-step 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
run() { tag=$1; shift
  MRK_RANK_JIT=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/${tag}_p1 -o s -- python bench.py $ARGS "$@" > $O/${tag}_p1.log 2>&1
  MRK_RANK_JIT=1 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${tag}_p2 -o s -- python bench.py $ARGS "$@" > $O/${tag}_p2.log 2>&1
  python tools/pmc_summary.py $O/${tag}_p1 $O/${tag}_p2 > $O/${tag}_summary.json
  python - $tag $O/${tag}_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
for k, v in d.items():
    if "rank_cells" in k:
        w = v.get("SQ_WAVES", {}).get("mean", 1) or 1
        print(sys.argv[1].ljust(12), k[:24], "waves", int(w), {c.replace("SQ_", ""): round(x.get("mean", 0) / w) for c, x in v.items() if c not in ("SQ_WAVES", "duration")})
PY
  find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
}
run base
run no_div_str --drop-features divers_genres,divers_actors,divers_tags
run no_profile --drop-features profile
run no_cross --drop-features profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity
run only_numbers --drop-features profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity,ctr,ctr_tag,ctr_genre,genre,title_length,position
