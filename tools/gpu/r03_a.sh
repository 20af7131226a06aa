# round 3, pass a: (1) can the GPU box install the real libraries? (2) new tests (big sort, id offsets,
# status merge), (3) c4 / c4x bench with the sample sort, kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_a
mkdir -p $O
( timeout 60 pip install --no-cache-dir lightgbm==4.6.0 'xgboost<3' onnxruntime > $O/pip_install.log 2>&1; echo "pip exit $?" >> $O/pip_install.log;
  timeout 20 pip download --no-cache-dir --no-deps -d /tmp/whl lightgbm==4.6.0 >> $O/pip_install.log 2>&1; echo "pip download exit $?" >> $O/pip_install.log;
  pip config list >> $O/pip_install.log 2>&1; ls /opt/wheelhouse 2>/dev/null | grep -i -E "lightgbm|xgboost|onnxruntime" >> $O/pip_install.log;
  python -c "import lightgbm" >> $O/pip_install.log 2>&1; python -c "import xgboost" >> $O/pip_install.log 2>&1; python -c "import onnxruntime" >> $O/pip_install.log 2>&1 ) 
tail -12 $O/pip_install.log
timeout 1200 python -m pytest tests/test_big_sort_gpu.py tests/test_serving_loop.py -m gpu -x -q > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
for w in c4 c4x; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/bench_$w.json 2> $O/bench_$w.log || tail -5 $O/bench_$w.log
  python - $w $O/bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', {k: round(v['avg_ms'] * v['launches_per_batch'], 3) for k, v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4x -o s -- python bench.py --workload c4x --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c4x.log 2>&1
head -25 $O/stats_c4x/*kernel_stats.csv 2>/dev/null | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o s -- python bench.py --workload c4 --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c4.log 2>&1
head -25 $O/stats_c4/*kernel_stats.csv 2>/dev/null | cut -c1-200
find $O -name "*kernel_trace.csv" -size +1M -delete
