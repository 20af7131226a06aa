#!/bin/bash
# Copies the judged summaries of a measurement pass (tools/gpu/r06_final.sh -> gpurun_out/<tag>/) into profiles/<tag>_*.
#   bash tools/collect_profiles.sh r06_g            (files named r06_g_*)
#   bash tools/collect_profiles.sh r06_an r06_zz     (the pass of GPU call r06_an under the tag r06_zz: bench.py quotes the PMC
#                                                     summary whose name sorts LAST, and r06_an sorts before r06_y)
O=gpurun_out/$1; T=${2:-$1}
for w in c2 c3 c4 c4x c5 c2_xgb100_d6; do [ -f $O/bench_$w.json ] && cp $O/bench_$w.json profiles/${T}_bench_$w.json; done
for w in c2 c3 c4x; do
  f=$(find $O/stats1_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${T}_${w}_kernel_stats.csv
  [ -f $O/pmc_${w}_summary.json ] && cp $O/pmc_${w}_summary.json profiles/${T}_pmc_${w}_summary.json
done
f=$(find $O/stats_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${T}_c2_bench_kernel_stats.csv
[ -f $O/pytest.log ] && cp $O/pytest.log profiles/${T}_pytest_gpu.log
[ -f $O/callers_mrk_rank.txt ] && cat $O/callers_mrk_rank.txt $O/callers_serve.txt $O/callers_mrk_rank_with_queue.txt > profiles/${T}_callers.txt 2>/dev/null
ls profiles | grep "^${T}_"
