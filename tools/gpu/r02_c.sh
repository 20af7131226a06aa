# round 2, third GPU pass: the restructured assembly (record in registers, second trip issued for all ops ahead) -
# parity first, then A/B of the variants on c2 (cache resident) and a 2 M-item out-of-cache c4x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_c}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
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
for v in "regs1" "regs0:MRK_JIT_REGS=0" "regs1w3:MRK_JIT_WAVES=3" "regs1w4:MRK_JIT_WAVES=4" "regs0w4:MRK_JIT_REGS=0 MRK_JIT_WAVES=4"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  env $envs timeout 600 python bench.py --workload c2 $Q > $O/c2_$name.json 2> $O/c2_$name.log || tail -3 $O/c2_$name.log
  show "c2 $name" $O/c2_$name.json
  env $envs timeout 900 python bench.py --workload c4x --clones 19 --items 2000000 $Q > $O/c4x_$name.json 2> $O/c4x_$name.log || tail -3 $O/c4x_$name.log
  show "c4x(2M of 2M) $name" $O/c4x_$name.json
done
env MRK_RANK_JIT=0 timeout 900 python bench.py --workload c4x --clones 19 --items 2000000 $Q > $O/c4x_nojit.json 2> $O/c4x_nojit.log
show "c4x(2M of 2M) nojit" $O/c4x_nojit.json
