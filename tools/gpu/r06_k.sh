# r06_k: the walked-past mark of the hash tables: parity, same-box A/B (MRK_JIT_DEFINES=MRK_TABLE_WALKED=0 = before) on c2 / c3 / c4x, phase clocks
O=gpurun_out/r06_k; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
export MRK_RANK_JIT=1
timeout 1200 python -m pytest tests -m gpu -x -q -k "rank_parity or known_answers or big or sharded or c4 or serving" 2>&1 | tail -5 | tee $O/pytest_k.log
for wl in c2 c3 c4x; do
for v in "MRK_JIT_DEFINES=MRK_TABLE_WALKED=0" "MRK_X=1" "MRK_JIT_DEFINES=MRK_TABLE_WALKED=0" "MRK_X=1"; do
  env $v timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/$wl.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()})"
done
done | tee $O/ab.txt
export MRK_LIB=$PWD/ab/clk/libmrk_hip.so MRK_BENCH_PHASE=1
timeout 300 python tools/phase_clocks.py c2 32 2>&1 | grep -v "$F" | tee $O/clk_c2_unloaded.txt
MRK_JIT_DEFINES=MRK_TABLE_WALKED=0 timeout 300 python tools/phase_clocks.py c2 32 2>&1 | grep -v "$F" | tee $O/clk_c2_unloaded_before.txt
