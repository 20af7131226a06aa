# r06_ak: waiting callers of the queue that look every MRK_SERVE_POLL_US instead of spinning out the tail; the overload guard on / off
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ak; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
{
  for v in "MRK_SERVE_POLL_US=4" "MRK_SERVE_POLL_US=0" "MRK_SERVE_POLL_US=4 MRK_SERVE_OVERLOAD_MS=0" "MRK_SERVE_POLL_US=2" "MRK_SERVE_POLL_US=8 MRK_SERVE_SLEEP_EXTRA_US=0" "MRK_SERVE_POLL_US=4"; do
    echo "== $v"
    env $v timeout 300 python tools/concurrent_bench.py --queue 16,64,80,128,256 600 2>&1 | grep -v "$F"
  done
  grep throttled /sys/fs/cgroup/cpu.stat
  echo "== mrk_serve_rank, defaults"
  timeout 300 python tools/concurrent_bench.py --serve 16,32,64,128 600 2>&1 | grep -v "$F"
} | tee $O/callers.txt
