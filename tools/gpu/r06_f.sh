# Round 6, call f: the front as it will stay (3 lanes, futex tickets) + the serving queue with the hardware-queue budget - callers
# table, the stress test with a writer, parity suites touched by the front / buffers, the default bench line.
O=gpurun_out/${TAG:-r06_f}; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 400 python tools/concurrent_bench.py 1,4,16,32,64,128,256 600 2>&1 | grep -v "$F" | tee $O/callers_mrk_rank.txt
timeout 400 python tools/concurrent_bench.py --serve 1,4,16,32,64,128 600 2>&1 | grep -v "$F" | tee $O/callers_serve.txt
MRK_RANK_JIT=1 timeout 1500 python -m pytest -x -q -m gpu tests/test_serving_loop.py tests/test_rank_parity.py tests/test_rank_one_gpu.py tests/test_model_weights_cpu.py tests/test_score_gpu.py \
  "tests/test_encoder_gpu.py::test_f32_attention_with_dead_key_blocks_before_the_first_live_key" "tests/test_encoder_gpu.py::test_c5_against_the_fp32_embedding_not_against_itself" -s > $O/pytest.log 2>&1
tail -3 $O/pytest.log; grep -n "requests/s with a writer\|C5 against" $O/pytest.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline()); print(round(d["value"] / 1e6, 1), "M items/s; e2e", d.get("e2e", {}).get("value"), "latency p50", d.get("latency", {}).get("p50_ms"))
PY
