# round 3, pass r: timeline of the end-to-end loop (kernels + copies per stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_r
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o s -- python bench.py --steps 2 --warmup 1 --batches-per-step 4 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0.25 > $O/tr.log 2>&1
ls -la $O/tr | head; python - <<'PY'
import csv, glob, collections
O = "gpurun_out/r03_r/tr"
k = list(csv.DictReader(open(glob.glob(O + "/*kernel_trace.csv")[0])))
m = list(csv.DictReader(open(glob.glob(O + "/*memory_copy_trace.csv")[0])))
ev = []
for r in k:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:26], r.get("Stream_Id", r.get("Queue_Id"))))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:22] , r.get("Stream_Id", "")))
ev.sort()
# the last 0.2 s are the e2e loop; print a window of ~3 ms from the middle of it
t_end = ev[-1][1]
win0 = t_end - 120_000_000
sel = [e for e in ev if e[0] >= win0][:90]
t0 = sel[0][0]
for s, e, n, st in sel:
    print(f"{(s - t0) / 1000:9.1f} us  +{(e - s) / 1000:7.1f}  stream {st:>4}  {n}")
# busy fraction of compute over the e2e window
import itertools
PY
find $O -name "*.csv" -size +8M -delete
