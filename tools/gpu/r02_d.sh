# round 2, fourth GPU pass: occupancy / workgroup-shape A/B of the restructured assembly kernel on c2 + phase clocks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_d}
O=gpurun_out/$TAG
mkdir -p $O
Q="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for v in "w4" "w5:MRK_JIT_WAVES=5" "w6:MRK_JIT_WAVES=6" "w8:MRK_JIT_WAVES=8" "w4t64:MRK_FUSED_THREADS=64" "w6t64:MRK_JIT_WAVES=6 MRK_FUSED_THREADS=64" "w4s3:BENCH_STREAMS=3" "w4s4:BENCH_STREAMS=4"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  extra=""; [[ "$envs" == BENCH_STREAMS=* ]] && extra="--streams ${envs#BENCH_STREAMS=}" && envs=""
  env $envs timeout 600 python bench.py --workload c2 $Q $extra > $O/c2_$name.json 2> $O/c2_$name.log || tail -3 $O/c2_$name.log
  show "c2 $name" $O/c2_$name.json
done
env MRK_JIT_WAVES=6 timeout 600 python bench.py --workload c3 $Q > $O/c3_w6.json 2> $O/c3_w6.log; show "c3 w6" $O/c3_w6.json
timeout 600 python bench.py --workload c3 $Q > $O/c3_w4.json 2> $O/c3_w4.log; show "c3 w4" $O/c3_w4.json
# where the cycles of a workgroup go (measurement build; the box is discarded afterwards)
MRK_DEFINES=MRK_PHASE_CLOCKS python -c "from metarank_amd import _native; _native.build(force=True)" > $O/phase_build.log 2>&1
MRK_DEFINES=MRK_PHASE_CLOCKS timeout 600 python tools/phase_clocks.py c2 > $O/phase_c2.txt 2>&1; cat $O/phase_c2.txt | tail -40
