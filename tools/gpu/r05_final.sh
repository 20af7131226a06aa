# Round 5, final measurement pass (the code is frozen before it: the PMC summaries carry this build's id and bench.py quotes them
# only for it): the whole GPU suite, the bench line of every workload, config 2 as written, kernel stats of the default bench
# command, PMC passes of c2 / c3 / c4x, the encoder's kernel stats at config 5's batch.
#   gpurun --timeout 3000 -- 'bash tools/gpu/r05_final.sh'
export TAG=${TAG:-r05_k} XGB=1
bash tools/gpu/r04_c.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_enc_f32 -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/encoder_bench_f32.json 2> $O/encoder_bench_f32.err
f=$(find $O/stats_enc_f32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/enc_f32_kernel_stats.csv
timeout 300 python tools/encoder_bench.py --quick --precision f16 --json > $O/encoder_bench_f16.json 2>/dev/null
echo "encoder f32: $(cat $O/encoder_bench_f32.json | tail -1)"; echo "encoder f16: $(cat $O/encoder_bench_f16.json | tail -1)"
timeout 400 python bench.py --streams 3 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_c2_streams3.json 2>/dev/null; python - $O/bench_c2_streams3.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("c2 with 3 streams:", round(d["value"] / 1e6, 1), "M items/s")
except Exception as e:
    print("streams3 failed", e)
PY
find $O -name "*kernel_trace.csv" -size +1M -delete
