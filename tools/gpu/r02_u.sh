# round 2: batched residual loads / full-tile fast path in the matrix products' epilogue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_u}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -3
for dma in 0 3; do
  env MRK_ENCODER_DMA=$dma timeout 900 python bench.py --workload c5 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --steps 5 --warmup 2 > $O/bench_c5_dma$dma.json 2> $O/bench_c5_dma$dma.log || tail -5 $O/bench_c5_dma$dma.log
  python - <<PY
import json
d = json.load(open("$O/bench_c5_dma$dma.json"))
print("c5 dma=$dma", round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', d['encoder']['ms_per_step'], d['encoder']['tflops'])
PY
done
for dma in 0 3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5_dma$dma -o s -- env MRK_ENCODER_DMA=$dma python bench.py --workload c5 --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c5_dma$dma.log 2>&1
head -8 $O/stats_c5_dma$dma/s_kernel_stats.csv | cut -c1-150
done
find $O -name "*kernel_trace.csv" -size +1M -delete
