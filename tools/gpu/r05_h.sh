# Round 5: concurrent callers - mrk_rank (batching front) against the serving queue (one slot per thread), closed loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_h}
mkdir -p $O
for t in 1 4 16 32; do
  timeout 200 python tools/concurrent_bench.py $t 400 100 2>/dev/null | tail -1
  timeout 200 python tools/concurrent_bench.py $t 400 100 --serve 2>/dev/null | grep -v "^  " | tail -1
done | tee $O/concurrent.txt
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"
