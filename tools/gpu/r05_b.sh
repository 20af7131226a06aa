# Round 5, second call: ABI 8 (contexts of one process, f32 encoder by default) - the whole GPU suite, the default bench line with
# the reference's latency protocol, config 5 in both precisions, and the matrix-core counters of the f32 forward pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_b}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/pytest_gpu.log | head -5
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python - $O <<'PY'
import json, sys
O = sys.argv[1]
for w in ("c2", "c5"):
    try:
        d = json.loads(open(f"{O}/bench_{w}.json").read().strip().splitlines()[-1])
        print(w, {k: d[k] for k in ("value", "ms_per_step")}, "roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "peak")})
        if w == "c2":
            print(" per_kernel", {k: round(v["frac"], 4) for k, v in d["roofline"]["per_kernel"].items()}, "whole", round(d["roofline"]["whole_path"]["frac"], 4))
            lat = d["latency"]; print(" p50", lat["p50_ms"], "serve", lat.get("serve_queue", {}).get("p50_ms"))
            for leg in ("mrk_rank", "mrk_serve_rank"):
                for row in lat["sweep"].get(leg, []):
                    print("  ", leg, row)
            print(" e2e", d.get("e2e", {}).get("frac_of_value"))
        else:
            e = d["encoder"]; print(" encoder", {k: e.get(k) for k in ("ms_per_step", "tflops", "frac_of_mfma_peak", "f32_batch", "fp16_vs_f32", "other_precision")}); print(" latency", d["latency"])
    except Exception as ex:
        print(w, "parse failed", ex); print(open(f"{O}/bench_{w}.err").read()[-1500:])
PY
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_enc_f32 -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/pmc_enc_f32.log 2>&1
python - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_enc_f32/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "gemm_f32_mfma" in n or "attention_f32" in n:
            agg[n.split("(")[0][-48:] + "/grid" + row.get("Grid_Size", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items()}
json.dump(out, open(f"{O}/pmc_enc_f32_summary.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k, {c: round(v) for c, v in d.items()})
PY
find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
