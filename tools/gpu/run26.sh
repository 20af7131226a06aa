cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_rank_parity.py -m gpu -x -q 2>&1 | tail -3
for th in 1 4 16 64; do
for cb in 1 0; do
MRK_RANK_COMBINE=$cb timeout 600 python tools/concurrent_bench.py $th 300 2>&1 | tail -1
done; done
