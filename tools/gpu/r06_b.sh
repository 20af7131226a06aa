# Round 6, call b: the pipelined scorer (qs_score_pipe_kernel) - parity of every scorer variant, then a same-box A/B of the
# scorer alone (tools/score_bench.py) by switch, then the bench line with and without it.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_b.sh'
O=gpurun_out/${TAG:-r06_b}; mkdir -p $O
timeout 900 python -m pytest tests/test_score_gpu.py -x -q -m gpu > $O/pytest_score.log 2>&1; tail -3 $O/pytest_score.log
for rows in 384000 100000 4000000; do
  for v in "MRK_QS_PIPE=0" "MRK_QS_PIPE=4" "MRK_QS_PIPE=2" "MRK_QS_PIPE=0 MRK_QS_SPLIT=8"; do
    echo "== rows=$rows $v: $(env $v timeout 120 python tools/score_bench.py $rows 24 lgbm 500 2>&1 | tail -1)"
  done
done | tee $O/score_ab.txt
for v in "MRK_QS_PIPE=0" "MRK_QS_PIPE=4" "MRK_QS_PIPE=0" "MRK_QS_PIPE=4"; do
  env $v timeout 300 python bench.py --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_c2_$v.json 2>$O/bench_err.txt
  python - $O/bench_c2_$v.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d.get("kernels") or {}
    print(sys.argv[2], round(d["value"] / 1e6, 1), "M items/s", d.get("ms_per_step"), k)
except Exception as e:
    print("bench failed", e)
PY
done | tee $O/bench_ab.txt
