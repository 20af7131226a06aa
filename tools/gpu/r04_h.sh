# Round 4, call H: (1) encoder GPU tests (the skinny f32 product of single-query calls); (2) wavefronts per tile of the scorer
# for few tiles (c4: 782 tiles; MRK_QS_SPLIT); (3) the c5 line with the auto-precision encoder.
#   gpurun --timeout 900 -- 'bash tools/gpu/r04_h.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_h}
mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_encoder.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_encoder.log
grep -E "passed|failed|error" $O/pytest_encoder.log | tail -3
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    l = d.get('latency') or {}
    print(sys.argv[1].ljust(22), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'p50', l.get('p50_ms') and round(l['p50_ms'], 4),
          'fp16 query p50', l.get('fp16_query_p50_ms'), (d.get('encoder') or {}).get('fp16_vs_f32'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
for rep in 1 2; do
  EXTRA="--workload c4 --latency-requests 0" run c4_split4_$rep MRK_QS_SPLIT=4
  EXTRA="--workload c4 --latency-requests 0" run c4_split8_$rep MRK_QS_SPLIT=8
  EXTRA="--workload c4 --latency-requests 0" run c4_split16_$rep MRK_QS_SPLIT=16
done
EXTRA="--workload c4 --latency-requests 0 --items 20000" run c4_20k_split4 MRK_QS_SPLIT=4
EXTRA="--workload c4 --latency-requests 0 --items 20000" run c4_20k_split8 MRK_QS_SPLIT=8
EXTRA="--workload c4 --latency-requests 0 --items 20000" run c4_20k_split16 MRK_QS_SPLIT=16
EXTRA="--workload c4 --latency-requests 0 --items 400000" run c4_400k_split4 MRK_QS_SPLIT=4
EXTRA="--workload c4 --latency-requests 0 --items 400000" run c4_400k_split8 MRK_QS_SPLIT=8
EXTRA="--workload c4 --latency-requests 0 --items 400000" run c4_400k_split16 MRK_QS_SPLIT=16
EXTRA="--workload c5 --latency-requests 60" run c5 A=1
} 2>&1 | tee $O/ab.txt
