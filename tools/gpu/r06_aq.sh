# r06_aq: the GPU suite again on the final tree (flakiness of the round's new tests), serving tests twice on top
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_aq; mkdir -p $O; : > $O/pytest_again.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/full.log 2>&1; echo "full suite rc=$?" >> $O/pytest_again.log
grep -E "passed|failed|error" $O/full.log | tail -3 >> $O/pytest_again.log
for i in 1 2; do
  timeout 600 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -q -p no:cacheprovider > $O/serving_$i.log 2>&1; echo "serving tests rc=$?" >> $O/pytest_again.log
  grep -E "passed|failed|error" $O/serving_$i.log | tail -2 >> $O/pytest_again.log
done
cat $O/pytest_again.log
