# r06_ab: staged sanity of the serving gangs (cheap; stops at the first failure), then the serving tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ab; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 240 python tools/gpu/serve_sanity.py 2>&1 | grep -v "$F" | tail -60 | tee $O/sanity.txt
if ! grep -q '^ok' $O/sanity.txt; then
  echo "== without the gang clock (MRK_SERVE_GANG_CLOCK=0)"
  MRK_SERVE_GANG_CLOCK=0 timeout 240 python tools/gpu/serve_sanity.py 2>&1 | grep -v "$F" | tail -60 | tee $O/sanity_noclock.txt
  exit 1
fi
timeout 600 python -m pytest tests/test_serving_loop.py tests/test_rank_one_gpu.py -m gpu -x -q -s 2>&1 | grep -v "$F" | tail -30 | tee $O/pytest_serving.log
