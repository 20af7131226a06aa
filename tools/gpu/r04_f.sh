# Round 4, call F: switch-only experiments on top of the signature-keyed kernels (no code variants: one build), same box.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r04_f.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_f}
mkdir -p $O
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    l = d.get('latency') or {}
    print(sys.argv[1].ljust(26), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()},
          'p50', l.get('p50_ms') and round(l['p50_ms'], 4), 'same-request', l.get('same_request_p50_ms') and round(l['same_request_p50_ms'], 4),
          'queue', (l.get('serve_queue') or {}).get('p50_ms') and round(l['serve_queue']['p50_ms'], 4),
          'queue same-request', ((l.get('serve_queue') or {}).get('same_request') or {}).get('p50_ms'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
EXTRA="--workload c2 --latency-requests 200" run c2_default A=1
EXTRA="--workload c2 --latency-requests 0 --streams 3" run c2_streams3 A=1
EXTRA="--workload c2 --latency-requests 0 --streams 4" run c2_streams4 A=1
EXTRA="--workload c2 --latency-requests 0" run c2_waves5 MRK_JIT_WAVES=5
EXTRA="--workload c2 --latency-requests 0" run c2_waves3 MRK_JIT_WAVES=3
EXTRA="--workload c2 --latency-requests 0" run c2_load50 MRK_TABLE_LOAD_PCT=50
EXTRA="--workload c2 --latency-requests 0" run c2_fused_score MRK_RANK_FUSED_SCORE=1
EXTRA="--workload c2 --latency-requests 0" run c2_budget56 MRK_JIT_DEFINES=MRK_PRE_GROUP_BUDGET=56
EXTRA="--workload c2 --latency-requests 0" run c2_budget96 MRK_JIT_DEFINES=MRK_PRE_GROUP_BUDGET=96
EXTRA="--workload c2 --latency-requests 0" run c2_default_again A=1
EXTRA="--workload c3 --latency-requests 0" run c3_default A=1
EXTRA="--workload c3 --latency-requests 0" run c3_slices3 MRK_FUSED_SLICES=3
EXTRA="--workload c3 --latency-requests 0" run c3_slices4 MRK_FUSED_SLICES=4
EXTRA="--workload c3 --latency-requests 0" run c3_waves5 MRK_JIT_WAVES=5
EXTRA="--workload c3 --latency-requests 0 --streams 3" run c3_streams3 A=1
EXTRA="--workload c5 --latency-requests 60" run c5 A=1
} 2>&1 | tee $O/ab.txt
python - $O/c2_default.json $O/c5.json <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print('c2 latency', json.dumps(d['latency'])[:1500])
try:
    e = json.load(open(sys.argv[2]))
    print('c5 encoder', e['encoder'].get('fp16_vs_f32'), e['encoder'].get('precision'), 'latency', e['latency'] and e['latency']['p50_ms'], e['roofline']['frac'])
except Exception as ex:
    print('c5', ex)
PY
