cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --streams 1 --steps 40 --warmup 4 --cpu-sample 16 --latency-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams=1', round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"
OUT=gpurun_out/pmc_h; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p1 -o s -- python bench.py --streams 1 --steps 3 --warmup 1 --cpu-sample 0 --latency-requests 0 > $OUT/p1.log 2>&1
python tools/pmc_summary.py $OUT/p1 > $OUT/s1.json
python - <<PY
import json
d=json.load(open("$OUT/s1.json"))
for k,v in d.items():
    if "fused_cells" in k: print({c: round(x.get("mean", 0)/7680,0) for c,x in v.items()})
PY
