#!/bin/bash
# encoder leg evidence: GPU tests of the encoder, kernel stats of tools/encoder_bench.py, bench.py --workload c5
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01_g
bash tools/gpu/enc.sh
cp gpurun_out/enc_prof/enc_kernel_stats.csv gpurun_out/r01_g/encoder_kernel_stats.csv
cp gpurun_out/enc_bench.log gpurun_out/r01_g/encoder_bench.log
bash tools/gpu/c5.sh
cp gpurun_out/c5/bench_c5.json gpurun_out/r01_g/bench_c5.json
