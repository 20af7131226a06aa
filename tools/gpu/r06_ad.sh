# r06_ad: the gang build's sanity, then (only if it is alive) the serving tests and the callers
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ad; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 120 python tools/gpu/serve_sanity.py 2>&1 | grep -v "$F" | tail -30 | tee $O/sanity.txt
grep -q '^ok' $O/sanity.txt || exit 1
timeout 600 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -x -q -s 2>&1 | grep -v "$F" | tail -30 | tee $O/pytest_serving.log
grep -q 'failed\|error' $O/pytest_serving.log && exit 1
{
  timeout 300 python tools/concurrent_bench.py --serve 1,4,16,32,64,128 600 2>&1 | grep -v "$F"
  timeout 300 python tools/concurrent_bench.py --queue 1,16,32,64,128,256 600 2>&1 | grep -v "$F"
  timeout 300 python tools/concurrent_bench.py 16,64,128 600 2>&1 | grep -v "$F"
} | tee $O/callers.txt
