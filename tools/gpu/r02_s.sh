# round 2: LDS-DMA matrix products of the encoder - tests, micro-benchmark, c5, kernel stats; A/B against MRK_ENCODER_DMA=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_s}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -5
for dma in 1 0; do
  env MRK_ENCODER_DMA=$dma timeout 600 python tools/encoder_bench.py > $O/encoder_bench_dma$dma.log 2>&1; echo "dma=$dma"; tail -5 $O/encoder_bench_dma$dma.log
  env MRK_ENCODER_DMA=$dma timeout 900 python bench.py --workload c5 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --steps 5 --warmup 2 > $O/bench_c5_dma$dma.json 2> $O/bench_c5_dma$dma.log || tail -5 $O/bench_c5_dma$dma.log
  python - <<PY
import json
d = json.load(open("$O/bench_c5_dma$dma.json"))
print("c5 dma=$dma", round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', d['encoder']['ms_per_step'], d['encoder']['tflops'])
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o s -- python bench.py --workload c5 --steps 2 --warmup 1 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 > $O/stats_c5.log 2>&1
head -14 $O/stats_c5/*/s_kernel_stats.csv 2>/dev/null | cut -c1-150 || find $O/stats_c5 -name "*kernel_stats.csv" | head
find $O -name "*kernel_trace.csv" -size +1M -delete
