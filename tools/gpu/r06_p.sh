# r06_p: tunables of the resident-table kernels, same box: searches per group (MRK_RT_Q), tokens fetched ahead per interacted_with field
# (MRK_IW_TOK: the benchmark's tag lists have 5), the fetch-ahead register budget per op group
O=gpurun_out/r06_p; mkdir -p $O
export MRK_RANK_JIT=1
for wl in c2 c4x; do
for v in "MRK_X=1" "MRK_JIT_DEFINES=MRK_RT_Q=2" "MRK_JIT_DEFINES=MRK_RT_Q=3" "MRK_JIT_DEFINES=MRK_IW_TOK=5" "MRK_JIT_DEFINES=MRK_IW_TOK=6" "MRK_JIT_DEFINES=MRK_IW_TOK=8" "MRK_JIT_DEFINES=MRK_PRE_GROUP_BUDGET=96" "MRK_JIT_DEFINES=MRK_PRE_GROUP_BUDGET=48" "MRK_X=1"; do
  env "$v" timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 100 --latency-sweep 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()}, 'p50', (d.get('latency') or {}).get('p50_ms'))"
done
done | tee $O/ab.txt
