# r06_ag: where does the process die after `--queue 256`?  (seen twice in five runs: "dumped core" after the last row)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ag; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
ulimit -c 0
for i in 1 2 3 4; do
  timeout 200 python -X faulthandler tools/concurrent_bench.py --queue 32,64,128,256 600 2>&1 | grep -v "$F" | tail -25
  echo "== run $i rc=${PIPESTATUS[0]}"
done | tee $O/crash.txt
