# Round 6, call c: the pipelined front of mrk_rank (lanes, device-resolved ids for combined batches, writer priority) - parity of
# everything that changed since r06_b, then closed-loop native callers by lane count, then the serving queue.
#   gpurun --timeout 2400 -- 'bash tools/gpu/r06_c.sh'
O=gpurun_out/${TAG:-r06_c}; mkdir -p $O
export MRK_RANK_JIT=1
timeout 1500 python -m pytest -x -q -m gpu tests/test_serving_loop.py tests/test_model_weights_cpu.py \
  tests/test_score_gpu.py "tests/test_encoder_gpu.py::test_f32_attention_with_dead_key_blocks_before_the_first_live_key" \
  "tests/test_encoder_gpu.py::test_c5_against_the_fp32_embedding_not_against_itself" -s > $O/pytest.log 2>&1; tail -5 $O/pytest.log; grep -n "requests/s with a writer\|C5 against" $O/pytest.log
unset MRK_RANK_JIT
for lanes in 1 2 4 8; do
  timeout 600 python tools/concurrent_bench.py --lanes $lanes 1,4,16,32,64,128,256 400 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" 
done | tee $O/callers_mrk_rank.txt
timeout 600 python tools/concurrent_bench.py --serve 1,4,16,32,64 400 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee $O/callers_serve.txt
MRK_RANK_COMBINE=0 timeout 600 python tools/concurrent_bench.py --lanes 8 1,16,64 400 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee $O/callers_no_combine.txt
