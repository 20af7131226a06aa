#!/bin/bash
# CPU-side preparation of the same-box A/B of the gated experiments (tools/gpu/r04_first.sh): one build of the library +
# the stock program's kernels per variant under ab/<name>/ (git-ignored; travels with gpurun).  Variants:
#   base  the default build
#   e1    -DMRK_PREPASS_WAVES: the pre-pass sections on different wavefronts (rank_device.hpp)
set -e
cd "$(dirname "$0")/.."
tools/ab_build.sh base
MRK_DEFINES="MRK_PREPASS_WAVES" MRK_JIT_DEFINES="MRK_PREPASS_WAVES=1" tools/ab_build.sh e1
python -c "from metarank_amd import _native; _native.build()" > /dev/null 2>&1   # the in-tree library: back to the default build
ls ab/*/
