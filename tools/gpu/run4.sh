timeout 900 python -m pytest tests/test_score_gpu.py -x -q 2>&1 | tail -3
MRK_QS_R=2 timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1
MRK_QS_R=2 timeout 300 python tools/score_bench.py 384000 24 lgbm 500 2>&1 | tail -1
MRK_QS_R=2 timeout 300 python tools/score_bench.py 1600000 24 lgbm 500 2>&1 | tail -1
bash tools/gpu/pmc_qs.sh 2>&1 | grep "qs_score"
