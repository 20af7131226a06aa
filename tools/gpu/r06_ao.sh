# r06_ao: what the driver runs at round end, on the final tree: smoke(), the default bench line (pmc_stale must be false: profiles/r06_zz_*)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ao; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?" | tee -a $O/smoke.txt
python - <<'PY' | tee -a $O/smoke.txt
import json
d=json.loads(open("gpurun_out/r06_ao/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["value"]/1e6,1), "M items/s; pmc_stale", r["pmc_stale"], "traffic", r["traffic"], "frac", round(r["frac"],4), "whole-path frac", round(r["by_whole_path_bytes"]["frac"],4), "valu", {k: round(v["frac"],3) for k,v in (r.get("valu_issue") or {}).items()})
c=d["latency"]["concurrent"]
for k,v in c.items():
    if isinstance(v,list): print(k, [(x["callers"], round(x["requests_per_s"]), round(x["p50_ms"],3), round(x["p99_ms"],3)) for x in v])
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
