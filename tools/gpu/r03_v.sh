# round 3, pass v: the f32 encoder mode (parity vs the fp32 graph; C5 against an independent embedding)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_v
mkdir -p $O
timeout 1500 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -s -k "f32 or fp32_embedding" > $O/pytest.log 2>&1; grep -E "passed|failed|error|Error|assert|C5 against" $O/pytest.log | tail -12
