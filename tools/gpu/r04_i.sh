# Round 4, call I: the highest-priority sort stream (MRK_SORT_PRIORITY) and the parallel host part of device-id batches
# (MRK_HOST_THREADS) - GPU suite, same-box A/B on c2 with the end-to-end loop, kernel stats of the default bench command.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r04_i.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04_i}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error" $O/pytest.log | tail -3
show() { python - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    l = d.get('latency') or {}
    e = d.get('e2e') or {}
    print(sys.argv[1].ljust(24), round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch',
          {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()}, 'e2e', e.get('value') and round(e['value']/1e6, 1), e.get('frac_of_value') and round(e['frac_of_value'], 3),
          e.get('host_ms_per_batch') and {k: round(v, 3) for k, v in e['host_ms_per_batch'].items()}, 'p50', l.get('p50_ms') and round(l['p50_ms'], 4), 'kernel', l.get('kernel_ms'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
B="--steps 10 --warmup 2 --cpu-sample 0 --e2e-seconds 1.5"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.log || tail -3 $O/$name.log; show $name $O/$name.json; }
{
for rep in 1 2; do
  EXTRA="--workload c2 --latency-requests 0" run c2_old_$rep MRK_SORT_PRIORITY=0 MRK_HOST_THREADS=1
  EXTRA="--workload c2 --latency-requests 0" run c2_prio_$rep MRK_SORT_PRIORITY=1 MRK_HOST_THREADS=1
  EXTRA="--workload c2 --latency-requests 0" run c2_prio_threads_$rep MRK_SORT_PRIORITY=1
done
EXTRA="--workload c2 --latency-requests 100" run c2_default A=1
EXTRA="--workload c3 --latency-requests 0" run c3_old MRK_SORT_PRIORITY=0 MRK_HOST_THREADS=1
EXTRA="--workload c3 --latency-requests 0" run c3_default A=1
} 2>&1 | tee $O/ab.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o s -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c2.log 2>&1
head -5 $O/stats_c2/s_kernel_stats.csv | cut -c1-160 | tee -a $O/ab.txt
find $O -name "*kernel_trace.csv" -size +1M -delete
