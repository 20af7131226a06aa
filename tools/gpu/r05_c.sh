# Round 5, third call: the sample sort with counts -> offsets folded into the classify pass (and bucket sizes), the specialised
# stand-alone pre-pass (config 4), how the f32 product fills the chip.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_c}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/pytest_gpu.log | head -5
timeout 600 python tools/sort_bench.py 100000 > $O/sort_bench.txt 2>&1; grep -v "^{" $O/sort_bench.txt | tail -12
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_sort -o s -- python tools/sort_bench.py 100000 > /dev/null 2>&1
f=$(find $O/trace_sort -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py $f 0 > $O/sort_by_grid.txt && grep "ss_\|sort" $O/sort_by_grid.txt | head -40
for v in 1 0; do MRK_JIT_PREPASS=$v timeout 400 python bench.py --workload c4 --cpu-sample 0 > $O/bench_c4_prepass$v.json 2> $O/bench_c4_prepass$v.err; python - $O/bench_c4_prepass$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1][-22:], round(d["value"] / 1e6, 1), "M items/s", round(d["ms_per_step"] / d["config"].get("device_batches_per_step", 128), 4), {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items()}, d.get("multi_gpu_projection", {}).get("speedup_ceiling"))
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_gemm -o s -- python tools/gemm_rounds.py > $O/gemm_rounds.log 2>&1
f=$(find $O/trace_gemm -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py $f 20 > $O/gemm_by_grid.txt && grep "gemm_f32_mfma" $O/gemm_by_grid.txt | head -40
for w in 2 3; do echo "f32 waves $w: $(MRK_ENCODER_F32_WAVES=$w timeout 300 python tools/encoder_bench.py --quick --precision f32 --json 2>/dev/null)"; done
find $O -name "*kernel_trace.csv" -size +1M -delete
