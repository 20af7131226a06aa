# Round 5: rocprofv3 --kernel-trace --stats of the bench command of every other workload (the default command's is r05_k_c2_bench_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_i}
mkdir -p $O
for w in c3 c4 c4x c5; do
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python bench.py --workload $w --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_$w.json 2> $O/bench_$w.err
  f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_bench_kernel_stats.csv && echo "== $w" && head -6 $O/${w}_bench_kernel_stats.csv | cut -c1-170
done
find $O -name "*kernel_trace.csv" -size +1M -delete
