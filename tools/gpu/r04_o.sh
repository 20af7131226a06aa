# Round 4, last call: the default-JIT-mode test after the stand-ins-first change, then the c2 counters again for the final
# build (FETCH / WRITE / wait passes + kernel stats; the summary carries this build's provenance).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_o}
mkdir -p $O
for i in 1 2 3; do timeout 120 python -m pytest tests/test_rank_one_gpu.py -m gpu -q -p no:cacheprovider -k "default_jit" > $O/pytest_$i.log 2>&1; echo "default_jit run $i: $(grep -E 'passed|failed' $O/pytest_$i.log | tail -1)"; done
PMC="--steps 3 --warmup 1 --batches-per-step 1 --streams 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
run() { w=$1; name=$2; shift 2; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${w}_$name -o s -- python bench.py --workload $w $PMC > $O/pmc_${w}_$name.log 2>&1; }
w=c2
run $w fetch FETCH_SIZE
run $w write WRITE_SIZE
run $w wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$w -o s -- python bench.py --workload $w $PMC > $O/stats1_$w.log 2>&1
python tools/pmc_summary.py $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_wait $O/stats1_$w > $O/pmc_${w}_summary.json
python - $O/pmc_c2_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d.get('_provenance'))
for k, v in d.items():
    if isinstance(v, dict) and ('rank_cells' in k or 'qs_score' in k):
        print(k[:40], {c: round(x.get('mean', x.get('avg_ns', 0)), 1) for c, x in v.items() if isinstance(x, dict)})
PY
find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
