cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ze
mkdir -p $O
timeout 900 python -m pytest tests/test_rank_parity.py tests/test_score_gpu.py tests/test_known_answers.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -2
TAG=ab3 VARIANTS="v5_nowrap v6_fixedsearch" WL="c2 c3" bash tools/gpu/ab.sh
