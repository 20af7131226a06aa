# Round 4, call D: the signature-keyed sinks (MRK_JIT_SIG) and the shared pre-pass of sliced batches (MRK_SLICE_PREPASS) -
# (1) the whole GPU suite, (2) same-box A/B by switch on c2 / c3 / c4x (two rounds), (3) instruction-mix PMC of c2 both ways.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04_d.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_d}
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  grep -E "passed|failed|error" $O/pytest.log | tail -3
fi
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1].ljust(22), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d.get('latency') and round(d['latency']['p50_ms'], 4))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
for rep in 1 2; do
  EXTRA="--workload c2 --latency-requests 200" run c2_sig0_$rep MRK_JIT_SIG=0
  EXTRA="--workload c2 --latency-requests 200" run c2_sig1_$rep MRK_JIT_SIG=1
  EXTRA="--workload c3 --latency-requests 0" run c3_sig0_old_$rep MRK_JIT_SIG=0 MRK_SLICE_PREPASS=0
  EXTRA="--workload c3 --latency-requests 0" run c3_sig1_old_$rep MRK_JIT_SIG=1 MRK_SLICE_PREPASS=0
  EXTRA="--workload c3 --latency-requests 0" run c3_sig1_shared_$rep MRK_JIT_SIG=1 MRK_SLICE_PREPASS=1
done
EXTRA="--workload c3 --latency-requests 0" run c3_sig1_shared_4slices MRK_JIT_SIG=1 MRK_SLICE_PREPASS=1 MRK_FUSED_SLICES=4
EXTRA="--workload c4x --latency-requests 0" run c4x_sig0 MRK_JIT_SIG=0
EXTRA="--workload c4x --latency-requests 0" run c4x_sig1 MRK_JIT_SIG=1
EXTRA="--workload c4 --latency-requests 0" run c4_sig0 MRK_JIT_SIG=0
EXTRA="--workload c4 --latency-requests 0" run c4_sig1 MRK_JIT_SIG=1
} 2>&1 | tee $O/ab.txt
# PMC: instruction mix and wait cycles of the c2 assembly kernel both ways (separate passes; no trace domains with --pmc)
ARGS="--workload c2 --streams 1 --batches-per-step 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
pmc() { name=$1; sig=$2; shift 2; MRK_JIT_SIG=$sig timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o s -- python bench.py $ARGS > $O/$name.log 2>&1; }
for sig in 0 1; do
pmc p1_sig$sig $sig SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pmc p2_sig$sig $sig SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH
python tools/pmc_summary.py $O/p1_sig$sig $O/p2_sig$sig > $O/pmc_c2_sig${sig}_summary.json
python - $O/pmc_c2_sig${sig}_summary.json <<'PY' | tee -a $O/ab.txt
import json, sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v, dict) and ("rank" in k or "qs_score" in k):
        w = v.get("SQ_WAVES", {}).get("mean", 1) or 1
        print(sys.argv[1][-22:], k[:40], "waves", w, {c: round(x.get("mean", 0) / w, 1) for c, x in v.items() if isinstance(x, dict) and c not in ("SQ_WAVES", "duration")})
PY
done
find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
