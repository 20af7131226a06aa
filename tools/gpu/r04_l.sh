# Round 4, call L: which HIP call takes 4 s in test_busy_serving_workgroups_do_not_stall_reallocations (intermittent)?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_l}
mkdir -p $O
for i in 1 2 3 4; do
  timeout 200 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/t$i -o s -- python -m pytest tests/test_rank_one_gpu.py -m gpu -q -p no:cacheprovider -k busy_serving > $O/t$i.log 2>&1
  echo "run $i: $(grep -E 'passed|failed' $O/t$i.log | tail -1)"
  python - $O/t$i <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        try:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
        except Exception:
            pass
rows.sort(reverse=True)
for d, fn, t in rows[:8]:
    print("   ", round(d / 1e6, 2), "ms", fn, t)
PY
  find $O/t$i -name "*.csv" -size +2M -delete
done 2>&1 | tee $O/summary.txt
