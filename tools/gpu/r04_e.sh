# Round 4, call E: (1) the whole GPU suite; (2) same-box A/B of the item-parallel kernel's LDS table copies (MRK_ITEMS_LDS)
# on c4 / c4x; (3) the config-5 line (encoder as roofline kernel, fp16-vs-f32 report, auto precision); (4) c2 default line.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04_e.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_e}
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  grep -E "passed|failed|error" $O/pytest.log | tail -3
fi
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1].ljust(22), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d.get('latency') and round(d['latency']['p50_ms'], 4),
          'roofline', d['roofline']['kernel'], d['roofline']['bound'], round(d['roofline']['frac'], 4), d.get('multi_gpu_projection') and d['multi_gpu_projection']['speedup_ceiling'],
          d.get('encoder') and d['encoder'].get('fp16_vs_f32'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
for rep in 1 2; do
  EXTRA="--workload c4x --latency-requests 0" run c4x_arena_$rep MRK_ITEMS_LDS=0
  EXTRA="--workload c4x --latency-requests 0" run c4x_lds_$rep MRK_ITEMS_LDS=1
  EXTRA="--workload c4 --latency-requests 0" run c4_arena_$rep MRK_ITEMS_LDS=0
  EXTRA="--workload c4 --latency-requests 0" run c4_lds_$rep MRK_ITEMS_LDS=1
done
EXTRA="--workload c5 --latency-requests 100" run c5 A=1
EXTRA="--workload c2 --latency-requests 200" run c2 A=1
EXTRA="--workload c3 --latency-requests 0" run c3 A=1
} 2>&1 | tee $O/ab.txt
python - $O/c2.json <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print('c2 serve queue', d['latency'].get('serve_queue'))
print('c2 provenance', d.get('provenance'))
PY
