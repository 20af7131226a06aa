# r06_aj: the overload guard of the serving queue (fewer CPUs than slots: overflow in numbers sends everybody through the front for
# MRK_SERVE_OVERLOAD_MS), against no guard, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_aj; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
{
  for v in 200 0 50 200; do
    echo "== MRK_SERVE_OVERLOAD_MS=$v"
    MRK_SERVE_OVERLOAD_MS=$v timeout 300 python tools/concurrent_bench.py --queue 32,64,80,128,256 600 2>&1 | grep -v "$F"
  done
  echo "== mrk_serve_rank, guard on"
  timeout 300 python tools/concurrent_bench.py --serve 64,128 600 2>&1 | grep -v "$F"
} | tee $O/callers.txt
timeout 600 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -x -q -s 2>&1 | grep -v "$F" | tail -14 | tee $O/pytest_serving.log
