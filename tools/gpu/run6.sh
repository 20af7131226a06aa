cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 c3 c4; do
timeout 900 python bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/bench_r1_d_$w.json 2> gpurun_out/bench_r1_d_$w.log || tail -5 gpurun_out/bench_r1_d_$w.log
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r1_d_$w.json"))
print("$w", round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['latency'], d['roofline']['kernel'], round(d['roofline']['frac'],4), d['cpu_baseline'] and round(d['cpu_baseline']['value']))
PY
done
MRK_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --workload c4 --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 2>&1 | tail -1 | cut -c1-300
MRK_BENCH_FORCE_DIST=1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 2>&1 | tail -1 | cut -c1-300
