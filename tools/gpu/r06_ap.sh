# r06_ap: kernel trace of 64 callers through mrk_rank with the queue started: the gangs' kernels (grid 8, resident ~ MRK_SERVE_LIFE_US)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ap; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python tools/concurrent_bench.py --queue 64 600 > $O/run.log 2>&1
grep "mrk_rank+queue" $O/run.log
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200; cp $f $O/queue_64_callers_kernel_stats.csv
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY' | tee $O/gangs.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rank_serve" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
g = sorted(set((r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows))
print(f"{len(rows)} launches of the serving kernel, grid x workgroup {g}, residency ms: min {min(d):.2f} median {sorted(d)[len(d)//2]:.2f} max {max(d):.2f}")
PY
find $O -name "*kernel_trace.csv" -size +1M -delete
