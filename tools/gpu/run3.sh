timeout 900 python -m pytest tests/test_score_gpu.py -x -q 2>&1 | tail -5
for K in 1 0; do
MRK_QS_KERNEL=$K MRK_QS_R=2 timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1 | sed "s/^/kernel=$K /"
done
MRK_QS_R=2 timeout 300 python tools/score_bench.py 384000 24 lgbm 500 2>&1 | tail -1
MRK_QS_R=2 timeout 300 python tools/score_bench.py 1600000 24 lgbm 500 2>&1 | tail -1
MISSING=per_node MRK_QS_R=2 timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1
