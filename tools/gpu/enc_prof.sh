#!/bin/bash
# kernel trace of the encoder micro-benchmark -> gpurun_out/enc_prof/enc_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/enc_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_prof -o enc -- python tools/encoder_bench.py > gpurun_out/enc_prof/bench.log 2>&1
tail -12 gpurun_out/enc_prof/bench.log
cut -c1-200 gpurun_out/enc_prof/enc_kernel_stats.csv | head -16
rm -f gpurun_out/enc_prof/enc_kernel_trace.csv gpurun_out/enc_prof/*.db
