timeout 900 python -m pytest tests/test_score_gpu.py -x -q 2>&1 | tail -5
for P in 1 0; do for R in 2 4; do for C in 2 8; do
MRK_QS_PIPE=$P MRK_QS_R=$R MRK_QS_CHUNK_KB=$C timeout 300 python tools/score_bench.py 409600 24 lgbm 500 2>&1 | tail -1 | sed "s/^/pipe=$P chunk=$C /"
done; done; done
MRK_QS_PIPE=1 MRK_QS_R=2 timeout 300 python tools/score_bench.py 393216 24 lgbm 500 2>&1 | tail -1
MRK_QS_PIPE=1 MRK_QS_R=2 timeout 300 python tools/score_bench.py 1600000 24 lgbm 500 2>&1 | tail -1
