# r06_al: where a combined batch of the front stalls while the queue's gangs are resident (MRK_FRONT_TRACE), and what kills the
# callers tool after its last row
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_al; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{
  for i in 1 2; do
    MRK_FRONT_TRACE=1 MRK_SERVE_OVERLOAD_MS=0 timeout 300 python tools/concurrent_bench.py --queue 128 600 2>&1 | grep -v "$F"
  done
  MRK_FRONT_TRACE=1 timeout 300 python tools/concurrent_bench.py 128 600 2>&1 | grep -v "$F"
  for i in 1 2 3 4; do
    timeout 300 python -X faulthandler tools/concurrent_bench.py --queue 64,128,256 600 2>&1 | grep -v "$F" | tail -40
    echo "== rc=${PIPESTATUS[0]}"
  done
} | tee $O/trace.txt
