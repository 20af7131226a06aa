# Round 6, call h: forced co-residency of the two hot kernels.  Alone, each fills a CU's LDS with its own workgroups (8 x 20 KB
# assembly, 8 x 18.5 KB scorer), so with two batches in flight the kernels mostly ALTERNATE and meet only in their tails.  Padding
# the dynamic LDS of both caps each at a share of the CU: do they then run side by side, and is the batch period shorter?
O=gpurun_out/${TAG:-r06_h}; mkdir -p $O
one() { s=$1; shift; env "$@" timeout 200 python bench.py --streams $s --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,1), 'M items/s', round(d['ms_per_device_batch'],4), 'ms/batch', {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"; }
for cfg in "A=0 Q=0 S=2" "A=40960 Q=0 S=2" "A=0 Q=38000 S=2" "A=40960 Q=38000 S=2" "A=32768 Q=30000 S=2" "A=53000 Q=26000 S=2" "A=26000 Q=53000 S=2" "A=40960 Q=38000 S=3" "A=40960 Q=38000 S=4" "A=32768 Q=30000 S=4" "A=0 Q=0 S=2"; do
  eval $cfg
  echo "asm_lds_min=$A qs_lds_min=$Q streams=$S: $(one $S MRK_FUSED_LDS_MIN=$A MRK_QS_LDS_MIN=$Q)"
done | tee $O/coresidency.txt
