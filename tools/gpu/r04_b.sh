# Round 4, second GPU call: the adopted default (pre-pass sections on different wavefronts + tables probed by aligned
# bucket; every other gated variant deleted).  (1) the whole GPU suite; (2) c2 / c3 reference on this box; (3) how much a
# lower table load is worth AT EQUAL RESIDENCY: MRK_TABLE_LOAD_PCT 50 / 37 / 25 grows the LDS tables (fewer resident
# workgroups), so each is compared with the default load padded to the same LDS (MRK_FUSED_LDS_MIN) - the question behind
# 4-byte table entries (the same LDS at half the load); (4) instruction-mix PMC passes of c2.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04_b.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_b
mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 0"
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1].ljust(18), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d.get('latency') and round(d['latency']['p50_ms'], 4))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
EXTRA="--workload c2 --latency-requests 300" run c2_default A=1
EXTRA="--workload c3 --latency-requests 0" run c3_default A=1
export EXTRA="--workload c2 --latency-requests 0"
run c2_load50 MRK_TABLE_LOAD_PCT=50
run c2_pad24k MRK_FUSED_LDS_MIN=24576
run c2_load37 MRK_TABLE_LOAD_PCT=37
run c2_pad28k MRK_FUSED_LDS_MIN=28672
run c2_load25 MRK_TABLE_LOAD_PCT=25
run c2_pad36k MRK_FUSED_LDS_MIN=36864
run c2_default_again A=1
} 2>&1 | tee $O/ab.txt
# PMC: instruction mix and wait cycles of the c2 assembly kernel (separate passes; no trace domains with --pmc)
ARGS="--workload c2 --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0"
pmc() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o s -- python bench.py $ARGS > $O/$name.log 2>&1; }
pmc p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pmc p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH
python tools/pmc_summary.py $O/p1 $O/p2 > $O/pmc_c2_summary.json
python - <<'PY' | tee -a gpurun_out/r04_b/ab.txt
import json
d=json.load(open("gpurun_out/r04_b/pmc_c2_summary.json"))
for k,v in d.items():
    if "rank" in k or "qs_score" in k:
        w = v.get("SQ_WAVES", {}).get("mean", 1) or 1
        print(k[:60], "waves", w, {c: round(x.get("mean", 0) / w, 1) for c, x in v.items() if c not in ("SQ_WAVES", "duration")}, v.get("duration"))
PY
