# r06_s: resident tables in the one-launch kernels (mrk_rank, the serving queue): the GPU suite, single-request latency and callers, same box
O=gpurun_out/r06_s; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
export MRK_RANK_JIT=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/pytest_gpu.log
for v in "MRK_JIT_DEFINES=MRK_FUSED_RT_MAX_SPLIT=0" "MRK_X=1" "MRK_JIT_DEFINES=MRK_FUSED_RT_MAX_SPLIT=0" "MRK_X=1"; do
  env "$v" timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 400 --latency-sweep 1000 --e2e-seconds 0 --concurrent-callers '64,128' 2>$O/c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d.get('latency') or {}
print('c2 $v', round(d['value']/1e6,1), 'M items/s p50', l.get('p50_ms'), 'serve', (l.get('serve') or {}).get('p50_ms'), 'sweep', {k: (v.get('rank') or {}).get('p50') for k, v in (l.get('sweep') or {}).get('sizes', {}).items()} if isinstance(l.get('sweep'), dict) else None, 'callers', l.get('concurrent'))"
done 2>&1 | tee $O/ab.txt
