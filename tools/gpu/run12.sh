cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kb in 0 10.5 11.5 12.25 13.25 14.5 16 20 26; do
MRK_QS_LDS_KB=$kb timeout 300 python tools/score_bench.py 384000 24 lgbm 500 2>&1 | tail -1 | sed "s/^/lds_kb=$kb /"
done
for rows in 200000 409600 800000; do
timeout 300 python tools/score_bench.py $rows 24 lgbm 500 2>&1 | tail -1 | sed "s/^/auto /"
MRK_QS_LDS_KB=0 timeout 300 python tools/score_bench.py $rows 24 lgbm 500 2>&1 | tail -1 | sed "s/^/natural /"
done
