# r06_r: resident tables without the queue in the split / sliced kernel (c3) against its staging sink, same box
O=gpurun_out/r06_r; mkdir -p $O
export MRK_RANK_JIT=1
timeout 900 python -m pytest tests -m gpu -x -q -k "rank_parity or known_answers or big or sharded or rank_one or serving" 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/pytest_k.log
for wl in c3 c2; do
for v in "MRK_JIT_DEFINES=MRK_FUSED_RT_MAX_SPLIT=0" "MRK_X=1" "MRK_JIT_DEFINES=MRK_FUSED_RT_MAX_SPLIT=0" "MRK_X=1"; do
  env "$v" timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 200 --latency-sweep 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()}, 'p50', (d.get('latency') or {}).get('p50_ms'))"
done
done | tee $O/ab.txt
