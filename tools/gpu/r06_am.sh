# r06_am: memory given back later while gangs are resident (the front's buffers regrow without waiting for them); the native stack
# of whatever kills the callers tool at exit
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_am; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{
  MRK_FRONT_TRACE=1 MRK_SERVE_OVERLOAD_MS=0 timeout 300 python tools/concurrent_bench.py --queue 128 600 2>&1 | grep -v "$F"
  for i in 1 2 3; do
    timeout 300 python tools/concurrent_bench.py --queue 64,80,128,256 600 2>&1 | grep -v "$F" | tail -60
    echo "== rc=${PIPESTATUS[0]}"
  done
  MRK_SERVE_OVERLOAD_MS=0 timeout 300 python tools/concurrent_bench.py --queue 64,80,128,256 600 2>&1 | grep -v "$F" | tail -60
} | tee $O/trace.txt
