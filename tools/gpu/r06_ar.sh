# r06_ar: the driver's launch form for the scaling bench with one rank (RCCL world of one), and a second default run, final tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ar; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_1.json 2> $O/bench_torchrun_1.log; echo "torchrun rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_ar/bench_torchrun_1.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e6,1), "M items/s", d["n_gpus"], d["steps"], d["warmup"], round(d["ms_per_step"],2), "ms/step; pmc_stale", d["roofline"]["pmc_stale"], "parallelism", d["config"]["parallelism"])
PY
