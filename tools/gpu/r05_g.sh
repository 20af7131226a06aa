# Round 5, after the final pass (no code change): smoke() as the driver runs it, the RCCL legs with a world of one, the driver's
# torchrun launch form with one rank.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_g}
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-200)"
for w in c2 c4; do
  MRK_BENCH_FORCE_DIST=1 timeout 400 python bench.py --workload $w --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_${w}_rccl_world1.json 2> $O/bench_${w}_rccl.err
  python - $O/bench_${w}_rccl_world1.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1][-28:], round(d["value"] / 1e6, 1), "M items/s", d["config"].get("parallelism"), d["scaling"])
except Exception as e:
    print("failed", e)
PY
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "torchrun rc=$?"
python - $O/bench_torchrun_n1.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("torchrun n=1:", round(d["value"] / 1e6, 1), "M items/s; traffic", r["traffic"], "pmc_stale", r["pmc_stale"], "valu_issue", {k: round(v["frac"], 3) for k, v in (r.get("valu_issue") or {}).items()})
except Exception as e:
    print("failed", e)
PY
