# r06_o: exact-length compact tables (6 KB for the benchmark's model: the fused kernel keeps its residency) + the hash tables' load factor
O=gpurun_out/r06_o; mkdir -p $O
export MRK_RANK_JIT=1
timeout 1500 python -m pytest tests -m gpu -x -q -k "rank_parity or known_answers or score or big or sharded or serving or rank_one or write_path or hgb" 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/pytest_k.log
for wl in c2 c3; do
for v in "MRK_LIB=$PWD/ab/base/libmrk_hip.so MRK_JIT_DEFINES=MRK_GET_K=1" "MRK_X=1" "MRK_TABLE_LOAD_PCT=85" "MRK_TABLE_LOAD_PCT=90" "MRK_LIB=$PWD/ab/base/libmrk_hip.so MRK_JIT_DEFINES=MRK_GET_K=1" "MRK_X=1" "MRK_TABLE_LOAD_PCT=85" "MRK_TABLE_LOAD_PCT=90"; do
  env $v timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 100 --latency-sweep 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl ${v:0:24}', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()}, 'p50', (d.get('latency') or {}).get('p50_ms'))"
done
done | tee $O/ab.txt
grep -h "lds\|LDS" $O/c2.log | head -5
