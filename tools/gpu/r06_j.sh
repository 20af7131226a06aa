# r06_j: the resident-table item-parallel kernel (mrk_jit_assemble_cells_rt): parity, same-box A/B on c4x / c4, phase clocks of c2 and c4x
O=gpurun_out/r06_j; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
export MRK_RANK_JIT=1
timeout 900 python -m pytest tests -m gpu -x -q -k "c4 or big or sharded or assembly_paths or items" 2>&1 | tail -5 | tee $O/pytest_k.log
unset MRK_RANK_JIT
B="--steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers ''"
for v in "MRK_ITEMS_RT=0" "MRK_ITEMS_RT=1" "MRK_ITEMS_RT=1 MRK_ITEMS_RT_THREADS=256" "MRK_ITEMS_RT=0"; do
  env $v timeout 600 python bench.py --workload c4x --steps 5 --warmup 2 2>$O/c4x.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4x $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()})"
done | tee $O/ab_c4x.txt
for v in "MRK_ITEMS_RT=0" "MRK_ITEMS_RT=1" "MRK_ITEMS_RT=1 MRK_ITEMS_RT_THREADS=512"; do
  env $v timeout 600 python bench.py --workload c4 --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 2>$O/c4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 $v', round(d['value']/1e6,1), 'M items/s', {k: round(x['avg_ms'],4) for k,x in d['kernels'].items()})"
done | tee $O/ab_c4.txt
# phase clocks (measurement build)
export MRK_LIB=$PWD/ab/clk/libmrk_hip.so MRK_BENCH_PHASE=1 MRK_RANK_JIT=1
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>&1 >/dev/null | grep "phase clocks\|per op" | tee $O/clk_c2.txt
timeout 300 python tools/phase_clocks.py c2 32 2>&1 | grep -v "$F" | tee $O/clk_c2_unloaded.txt
for v in 0 1; do
MRK_ITEMS_RT=$v MRK_PHASE_KERNEL=items timeout 600 python bench.py --workload c4x --steps 3 --warmup 1 2>&1 >/dev/null | grep "phase clocks\|per op" | tee $O/clk_c4x_rt$v.txt
done
