# r06_af: sleep calibration of the queue's callers (MRK_SERVE_SLEEP_EXTRA_US) and mrk_rank's lock-free way into the queue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_af; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 600 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -x -q -s 2>&1 | grep -v "$F" | tail -12 | tee $O/pytest_serving.log
{
  for v in 8 0 16; do
    echo "== MRK_SERVE_SLEEP_EXTRA_US=$v"
    MRK_SERVE_SLEEP_EXTRA_US=$v timeout 300 python tools/concurrent_bench.py --serve 32,64,128 600 2>&1 | grep -v "$F"
    MRK_SERVE_SLEEP_EXTRA_US=$v timeout 300 python tools/concurrent_bench.py --queue 32,64,128,256 600 2>&1 | grep -v "$F"
  done
  echo "== MRK_SERVE_SPIN_CALLERS=4"
  MRK_SERVE_SPIN_CALLERS=4 timeout 300 python tools/concurrent_bench.py --queue 8,16,32,64,128 600 2>&1 | grep -v "$F"
  grep throttled /sys/fs/cgroup/cpu.stat
} | tee $O/callers.txt
