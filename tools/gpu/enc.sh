#!/bin/bash
# encoder leg: GPU tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/enc_test.log
cat gpurun_out/enc_test.log
