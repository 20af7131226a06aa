# round 2, second GPU pass: record layout v2 (128-byte aligned records, inline tokens) - the suite, the four workloads,
# HBM-side traffic of the assembly kernels on c2 (cache resident) and c4x (8 M items, out of every cache)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_b}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for w in c2 c3 c4 c4x; do
timeout 900 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.log || tail -5 $O/bench_$w.log
python - <<PY
import json
d=json.load(open("$O/bench_$w.json"))
print("$w", round(d['value']/1e6,1),'M items/s', round(d['ms_per_device_batch'],3),'ms/batch', {k:round(v['avg_ms']*v['launches_per_batch'],3) for k,v in d['kernels'].items()}, d['latency'], d['roofline']['kernel'], round(d['roofline']['frac'],4), 'e2e', d['e2e'] and round(d['e2e']['value']/1e6,1))
PY
done
PMC="--steps 3 --warmup 1 --batches-per-step 1 --streams 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
run() { w=$1; name=$2; shift 2; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${w}_$name -o s -- python bench.py --workload $w $PMC > $O/pmc_${w}_$name.log 2>&1; }
for w in c2 c4x; do
run $w fetch FETCH_SIZE
run $w write WRITE_SIZE
run $w tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run $w wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python bench.py --workload $w $PMC > $O/stats_$w.log 2>&1
python tools/pmc_summary.py $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_tcc $O/pmc_${w}_wait $O/stats_$w > $O/pmc_${w}_summary.json
python - <<PY
import json
d=json.load(open("$O/pmc_${w}_summary.json"))
for k,v in d.items():
    if any(x in k for x in ("rank", "assemble", "qs_score", "prepass", "resolve")):
        print("$w", k[:48], {c: round(x.get("mean", x.get("avg_ns",0)),1) for c,x in v.items()})
PY
done
# keep the CSVs small: the per-launch counter files are what the summaries were made from
find $O -name "*_counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
