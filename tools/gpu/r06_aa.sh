# r06_aa: the serving queue in gangs (8 resident workgroups per kernel, one stream per gang, 64 slots) - its tests, then
# closed-loop native callers through mrk_serve_rank, through mrk_rank with the queue started, and through mrk_rank's front alone
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_aa; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 900 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -x -q -s 2>&1 | grep -v "$F" | tail -40 | tee $O/pytest_serving.log
{
  timeout 400 python tools/concurrent_bench.py --serve 1,4,16,32,64,128 600 2>&1 | grep -v "$F"
  timeout 400 python tools/concurrent_bench.py --queue 1,16,32,64,128,256 600 2>&1 | grep -v "$F"
  timeout 400 python tools/concurrent_bench.py 16,64,128 600 2>&1 | grep -v "$F"
} | tee $O/callers.txt
