# r06_q: new defaults (MRK_IW_TOK 5, MRK_PRE_GROUP_BUDGET 48) against the old ones, wavefronts per SIMD of the item-parallel kernel
O=gpurun_out/r06_q; mkdir -p $O
export MRK_RANK_JIT=1
timeout 900 python -m pytest tests -m gpu -x -q -k "rank_parity or known_answers or big or sharded" 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/pytest_k.log
for wl in c2 c3 c4x; do
for v in "MRK_JIT_DEFINES=MRK_IW_TOK=4 MRK_PRE_GROUP_BUDGET=72" "MRK_X=1" "MRK_JIT_DEFINES=MRK_IW_TOK=4 MRK_PRE_GROUP_BUDGET=72" "MRK_X=1"; do
  env "$v" timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 100 --latency-sweep 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()}, 'p50', (d.get('latency') or {}).get('p50_ms'))"
done
done | tee $O/ab.txt
for v in "MRK_JIT_WAVES=5 MRK_ITEMS_RT_THREADS=256" "MRK_JIT_WAVES=6 MRK_ITEMS_RT_THREADS=256" "MRK_JIT_WAVES=3 MRK_ITEMS_RT_THREADS=256" "MRK_ITEMS_RT_THREADS=256" "MRK_X=1"; do
  env $v timeout 600 python bench.py --workload c4x --steps 5 --warmup 2 2>$O/c4x.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4x $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()})"
done | tee -a $O/ab.txt
