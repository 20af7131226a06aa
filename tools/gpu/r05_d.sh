# Round 5, fourth call: the f32 128 x 128 product on the 32 x 32 x 2 form (A/B), the native host of the end-to-end loop, config 4 on the final sort
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_d}
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_big_sort_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest_some.log 2>&1; echo "encoder + sort tests rc=$? $(grep -E 'passed|failed' $O/pytest_some.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/pytest_some.log | head -5
for v in 0 1 0 1; do echo "f32 mfma32=$v: $(MRK_ENCODER_F32_MFMA32=$v timeout 300 python tools/encoder_bench.py --quick --precision f32 --json 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['c5_batch_3840'], d['single_query']['p50_ms'])")"; done
MRK_ENCODER_F32_MFMA32=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_enc32 -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/stats_enc32.log 2>&1
f=$(find $O/stats_enc32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/enc_f32_mfma32_kernel_stats.csv && head -6 $O/enc_f32_mfma32_kernel_stats.csv | cut -c1-180
timeout 600 python bench.py --latency-sweep 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["e2e"]
print("value", round(d["value"] / 1e6, 1), "python e2e", round(e["frac_of_value"], 3), e["host_ms_per_batch"])
for k, v in (e.get("native_driver") or {}).items():
    print(" native", k, round(v["value"] / 1e6, 1), round(v["frac_of_value"], 3), {a: round(b, 4) for a, b in v["host_ms_per_batch"].items()})
PY
timeout 400 python bench.py --workload c4 --cpu-sample 0 > $O/bench_c4.json 2> $O/bench_c4.err; python - $O/bench_c4.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c4", round(d["value"] / 1e6, 1), "M items/s", {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items()}, d.get("multi_gpu_projection", {}).get("speedup_ceiling"))
PY
find $O -name "*kernel_trace.csv" -size +1M -delete
