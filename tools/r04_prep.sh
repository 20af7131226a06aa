#!/bin/bash
# CPU-side preparation of the same-box A/B of the gated experiments (tools/gpu/r04_first.sh): one build of the library +
# the stock program's kernels per variant under ab/<name>/ (git-ignored; travels with gpurun), the variant's JIT defines in
# ab/<name>/jit_defines.  ~5 min per variant on this container.  Variants (rank_device.hpp):
#   base   the default build
#   e1     MRK_PREPASS_WAVES: the pre-pass sections on different wavefronts
#   e4     MRK_GET_PAIR: two lookups' home windows per LDS trip in the per-item phase
#   e4w2   the same with 2-entry windows (the registers of one 4-entry window)
#   e5     MRK_LEAN_GET: the lookup's bookkeeping per window instead of per entry (-8 % static instructions)
#   e45    e4 + e5;  e145: all three (a build the parity suites run over)
#   e10    MRK_TABLE_BUCKETS: the tables probed by aligned bucket (-34 % static instructions); e10p: + pairs; e1_10p: + e1 (parity suites too)
#   e11    MRK_TABLE_2CHOICE: two home buckets per key (the slowest of 64 lookups: ~3 trips instead of ~8 at 75 % load); e1_11: + e1 (parity too)
#   pc_*   base / e1 with MRK_PHASE_CLOCKS (clock64() stamps at the phase boundaries)
set -e
cd "$(dirname "$0")/.."
variant() {  # name, defines...
  local name=$1; shift
  local aot="" jit=""
  for d in "$@"; do aot="$aot $d"; jit="$jit $d"; done
  MRK_DEFINES="${aot# }" MRK_JIT_DEFINES="${jit# }" tools/ab_build.sh $name
  echo "${jit# }" > ab/$name/jit_defines
}
variant base
variant e1 MRK_PREPASS_WAVES=1
variant e4 MRK_GET_PAIR=1
variant e4w2 MRK_GET_PAIR=1 MRK_PROBE_W=2
variant e5 MRK_LEAN_GET=1
variant e45 MRK_GET_PAIR=1 MRK_LEAN_GET=1
variant e145 MRK_PREPASS_WAVES=1 MRK_GET_PAIR=1 MRK_LEAN_GET=1
variant e10 MRK_TABLE_BUCKETS=1
variant e10p MRK_TABLE_BUCKETS=1 MRK_GET_PAIR=1
variant e1_10 MRK_PREPASS_WAVES=1 MRK_TABLE_BUCKETS=1
variant e1_10p MRK_PREPASS_WAVES=1 MRK_TABLE_BUCKETS=1 MRK_GET_PAIR=1
variant e11 MRK_TABLE_BUCKETS=1 MRK_TABLE_2CHOICE=1
variant e1_11 MRK_PREPASS_WAVES=1 MRK_TABLE_BUCKETS=1 MRK_TABLE_2CHOICE=1
if [ "$1" != "nopc" ]; then   # measurement builds: tools/phase_clocks.py (cycles per phase of an unloaded request)
  variant pc_base MRK_PHASE_CLOCKS=1
  variant pc_e1 MRK_PHASE_CLOCKS=1 MRK_PREPASS_WAVES=1
fi
python -c "from metarank_amd import _native; _native.build()" > /dev/null 2>&1   # the in-tree library: back to the default build
ls ab/*/
