# Round 4, first GPU call: the gated experiments written at the end of round 3 (tools/r04_prep.sh builds ab/<variant> on
# the CPU side first).  (1) the parity suites over the e145 build (all experiments; generic AND specialised kernels);
# (ab/ is listed in .gpurunignore between such calls: take the line out first.)
# (2) same-box A/B of every variant against base: c2, c3 and the single-request latency.   gpurun --timeout 2100 -- 'bash tools/gpu/r04_first.sh'  (about 30 GPU-minutes; VARIANTS="..." trims the A/B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_first
mkdir -p $O
# e145 / e1_10p / e1_11 hold every experiment of the linear-probing, the bucketed and the two-choice tables: three parity runs cover them all
# (a failure is narrowed down with the single-experiment builds afterwards)
for pv in e1_11 e1_10p e145; do
  MRK_LIB=$PWD/ab/$pv/libmrk_hip.so MRK_JIT_DEFINES="$(cat ab/$pv/jit_defines)" timeout 400 python -m pytest tests/test_known_answers.py tests/test_rank_parity.py \
    tests/test_rank_one_gpu.py tests/test_big_sort_gpu.py tests/test_serving_loop.py -m gpu -x -q -p no:cacheprovider > $O/pytest_$pv.log 2>&1
  echo "$pv parity rc=$?" | tee -a $O/pytest_$pv.log
  grep -E "passed|failed|error" $O/pytest_$pv.log | tail -3
done
for rep in 1 2; do for v in ${VARIANTS:-base e1 e5 e45 e10 e10p e11 e1_10 e1_10p e1_11 e145}; do for w in c2 c3; do
  [ $rep = 2 ] && [ $w = c3 ] && continue   # c3 once, c2 twice
  D="$(cat ab/$v/jit_defines)"
  MRK_JIT_DEFINES="$D" MRK_LIB=$PWD/ab/$v/libmrk_hip.so timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --cpu-sample 0 \
    --latency-requests $([ $w = c2 ] && echo 300 || echo 0) --e2e-seconds 0 > $O/${v}_${w}_$rep.json 2> $O/${v}_${w}_$rep.log || tail -3 $O/${v}_${w}_$rep.log
  python - $v $w $rep $O/${v}_${w}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[4]))
    print(sys.argv[1].ljust(6), sys.argv[2], sys.argv[3], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, d['latency'] and round(d['latency']['p50_ms'], 4))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
done; done; done 2>&1 | tee $O/ab.txt
# the out-of-cache gather (4 M candidates over an 8 M-item table): the lookups of MRK_GET_PAIR are in its kernel too
for v in base e45 e10p e11; do
  D="$(cat ab/$v/jit_defines)"
  MRK_JIT_DEFINES="$D" MRK_LIB=$PWD/ab/$v/libmrk_hip.so timeout 300 python bench.py --workload c4x --steps 5 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 \
    > $O/${v}_c4x.json 2> $O/${v}_c4x.log || tail -3 $O/${v}_c4x.log
  python -c "
import json,sys
d=json.load(open('$O/${v}_c4x.json')); print('$v'.ljust(6),'c4x',round(d['value']/1e6,1),'M items/s',{k: round(x['avg_ms']*x['launches_per_batch'],3) for k,x in d['kernels'].items()})" 2>&1 | tail -1
done 2>&1 | tee -a $O/ab.txt

# the UNFUSED path (pre-pass kernel + item-parallel assembly: tables through L2) under the same experiments: is the fused kernel still ahead?
for v in base e1_10p e1_11; do
  MRK_RANK_FUSED=0 MRK_JIT_DEFINES="$(cat ab/$v/jit_defines)" MRK_LIB=$PWD/ab/$v/libmrk_hip.so timeout 300 python bench.py --workload c2 --steps 10 --warmup 2 --cpu-sample 0 \
    --latency-requests 0 --e2e-seconds 0 > $O/${v}_c2_unfused.json 2> $O/${v}_c2_unfused.log || tail -3 $O/${v}_c2_unfused.log
  python -c "
import json
d=json.load(open('$O/${v}_c2_unfused.json')); print('$v'.ljust(6),'c2 unfused',round(d['value']/1e6,1),'M items/s',{k: round(x['avg_ms']*x['launches_per_batch'],3) for k,x in d['kernels'].items()})" 2>&1 | tail -1
done 2>&1 | tee -a $O/ab.txt
# where an unloaded request's cycles go, default pre-pass vs sections on different wavefronts (32 requests: a workgroup per CU)
for v in pc_base pc_e1; do
  MRK_JIT_DEFINES="$(cat ab/$v/jit_defines)" MRK_LIB=$PWD/ab/$v/libmrk_hip.so MRK_RANK_JIT=1 MRK_FUSED_SPLIT=1 timeout 200 python tools/phase_clocks.py c2 32 > $O/phase_$v.txt 2>&1
  echo "== $v"; head -9 $O/phase_$v.txt
done
