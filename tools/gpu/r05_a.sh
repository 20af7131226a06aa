# Round 5, first call: the f32 encoder on the matrix cores (v_mfma_f32_16x16x4_f32) - parity suite, rates, kernel stats -
# then the whole GPU suite and the default bench line as this round's starting point.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_a}
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest_encoder.log 2>&1; echo "encoder tests rc=$? $(grep -E 'passed|failed' $O/pytest_encoder.log | tail -1)"
for p in f32 f16; do timeout 300 python tools/encoder_bench.py --quick --precision $p --json > $O/encoder_bench_$p.json 2> $O/encoder_bench_$p.err; echo "$p: $(cat $O/encoder_bench_$p.json)"; done
MRK_ENCODER_F32_MFMA=0 timeout 300 python tools/encoder_bench.py --quick --precision f32 --json > $O/encoder_bench_f32_valu.json 2>/dev/null; echo "f32 valu: $(cat $O/encoder_bench_f32_valu.json)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_enc_f32 -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/stats_enc_f32.log 2>&1
f=$(find $O/stats_enc_f32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/enc_f32_kernel_stats.csv && head -12 $O/enc_f32_kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -size +1M -delete
if [ -z "$SKIP_SUITE" ]; then
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; python - $O/bench_c2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d.get("roofline", {}).get("frac"), d.get("latency", {}))
except Exception as e:
    print("bench parse failed", e)
PY
fi
