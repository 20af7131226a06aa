# Round 4, measurement pass: the whole -m gpu suite, the bench line of every workload, kernel stats of the default bench
# command, PMC passes of c2 / c3 / c4x (summaries carry the provenance of the run they were taken on).
#   gpurun --timeout 1500 -- 'TAG=r04_c bash tools/gpu/r04_c.sh'            (first pass of the round: c2 / c3 only)
#   gpurun --timeout 1800 -- 'TAG=r04_g XGB=1 bash tools/gpu/r04_c.sh'      (final pass: everything)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04_c}
O=gpurun_out/$TAG
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  grep -E "passed|failed|error" $O/pytest.log | tail -3
fi
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()},
          'lat', d['latency'] and round(d['latency']['p50_ms'], 3), 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'], 4), 'e2e', d['e2e'] and round(d['e2e']['value']/1e6, 1),
          'cpu', d['cpu_baseline'] and round(d['cpu_baseline']['value']), d['config'].get('scorer'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.log || tail -5 $O/bench_c2.log
show c2 $O/bench_c2.json
for w in ${WORKLOADS:-c3 c4 c4x c5}; do
  timeout 600 python bench.py --workload $w --cpu-sample 0 > $O/bench_$w.json 2> $O/bench_$w.log || tail -5 $O/bench_$w.log
  show $w $O/bench_$w.json
done
if [ -n "$E2E2" ]; then   # the serving loop driven by two host threads (each its own batches)
  timeout 600 python bench.py --cpu-sample 0 --latency-requests 0 --e2e-threads 2 > $O/bench_c2_e2e2.json 2> $O/bench_c2_e2e2.log; show "c2, 2 host threads" $O/bench_c2_e2e2.json
  timeout 600 python bench.py --cpu-sample 0 --latency-requests 0 --e2e-threads 2 --e2e-batches 3 > $O/bench_c2_e2e2b3.json 2> $O/bench_c2_e2e2b3.log; show "c2, 2 host threads x 3 batches" $O/bench_c2_e2e2b3.json
  timeout 600 env MRK_JIT_SIG=0 python bench.py --workload c3 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_c3_sig0.json 2> $O/bench_c3_sig0.log; show "c3, program-only kernels (same box)" $O/bench_c3_sig0.json
fi
if [ -n "$XGB" ]; then
  timeout 600 python bench.py --backend xgboost --trees 100 --depth 6 --cpu-sample 0 > $O/bench_c2_xgb100_d6.json 2> $O/bench_c2_xgb100_d6.log; show "config 2 as written (xgboost 100 x depth 6)" $O/bench_c2_xgb100_d6.json
fi
# kernel stats of the default bench command (2 streams: what the driver's line is made of)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o s -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c2.log 2>&1
PMC="--steps 3 --warmup 1 --batches-per-step 1 --streams 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
run() { w=$1; name=$2; shift 2; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${w}_$name -o s -- python bench.py --workload $w $PMC > $O/pmc_${w}_$name.log 2>&1; }
for w in ${PMC_WORKLOADS:-c2 c3 c4x}; do
run $w fetch FETCH_SIZE
run $w write WRITE_SIZE
run $w wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$w -o s -- python bench.py --workload $w $PMC > $O/stats1_$w.log 2>&1
python tools/pmc_summary.py $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_wait $O/stats1_$w > $O/pmc_${w}_summary.json
python - <<PY
import json
d=json.load(open("$O/pmc_${w}_summary.json"))
for k,v in d.items():
    if isinstance(v, dict) and any(x in k for x in ("rank_cells", "assemble_cells", "qs_score", "prepass", "resolve", "ss_", "sort")):
        print("$w", k[:40], {c: round(x.get("mean", x.get("avg_ns",0)),1) for c,x in v.items() if isinstance(x, dict)})
PY
done
find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
