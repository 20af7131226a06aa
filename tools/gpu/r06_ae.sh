# r06_ae: callers of the serving queue that sleep through the device's time (MRK_SERVE_SPIN_CALLERS), on a box with a CPU quota
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ae; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
{
  echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc: $(nproc) cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
  cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
  for v in 8 0 1000; do
    echo "== MRK_SERVE_SPIN_CALLERS=$v"
    MRK_SERVE_SPIN_CALLERS=$v timeout 300 python tools/concurrent_bench.py --serve 16,32,64,128 600 2>&1 | grep -v "$F"
  done
  cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
  echo "== mrk_rank with the queue started"
  timeout 300 python tools/concurrent_bench.py --queue 1,16,64,128,256 600 2>&1 | grep -v "$F"
} | tee $O/callers.txt
