# round 2: epilogue with one f32->f16 rounding everywhere; tile variants again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02_v}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|rror" $O/pytest.log | tail -3
for dma in 0 1 2 4; do
  env MRK_ENCODER_DMA=$dma timeout 900 python bench.py --workload c5 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --steps 5 --warmup 2 > $O/bench_c5_dma$dma.json 2> $O/bench_c5_dma$dma.log || tail -5 $O/bench_c5_dma$dma.log
  python - <<PY
import json
d = json.load(open("$O/bench_c5_dma$dma.json"))
print("c5 dma=$dma", round(d['value']/1e6, 1), 'M items/s', round(d['ms_per_device_batch'], 3), 'ms/batch', d['encoder']['ms_per_step'], d['encoder']['tflops'])
PY
done
env MRK_ENCODER_DMA=1 timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "batch_it_travels or minilm_shape or packed" > $O/pytest_dma1.log 2>&1; grep -E "passed|failed|rror" $O/pytest_dma1.log | tail -3
timeout 600 python tools/encoder_bench.py > $O/encoder_bench.log 2>&1; tail -9 $O/encoder_bench.log
