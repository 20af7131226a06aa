cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_big_sort_gpu.py --deselect tests/test_rank_one_gpu.py -k "not (test_known or test_score_gpu or test_encoder)" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
bash tools/gpu/r03_f.sh
