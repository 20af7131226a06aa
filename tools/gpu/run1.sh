set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_score_gpu.py -x -q 2>&1 | tail -15
for R in 2 4; do MRK_QS_R=$R timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -2; done
MRK_SCORER=walk timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1
MISSING=per_node MRK_QS_R=2 timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1
