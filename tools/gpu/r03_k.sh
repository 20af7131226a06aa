# round 3, pass k: LDS counters of the c2 assembly kernel (is the LDS pipe the limiter?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_k
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "LDS|TA_|TCP_" | head -60 > $O/counters.txt; wc -l $O/counters.txt
ARGS="--streams 1 --steps 2 --warmup 1 --batches-per-step 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
run() { tag=$1; shift
  MRK_RANK_JIT=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o s -- python bench.py $ARGS > $O/$tag.log 2>&1 || tail -3 $O/$tag.log
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run p2 SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY
python tools/pmc_summary.py $O/p1 $O/p2 > $O/summary.json
python - $O/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "rank_cells" in k or "qs_score" in k:
        w = v.get("SQ_WAVES", {}).get("mean", 1) or 1
        print(k[:24], "waves", int(w), {c.replace("SQ_", ""): round(x.get("mean", 0) / w, 1) for c, x in v.items() if c not in ("SQ_WAVES", "duration")})
PY
grep -i "lds" $O/counters.txt | head -30
find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
