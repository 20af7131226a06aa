# Round 6, call d: after geometric regrowth of the lanes' buffers - callers by lane count; the serving queue with more hardware
# queues than slots (GPU_MAX_HW_QUEUES) to test the cause of its collapse; the writer stress test.
O=gpurun_out/${TAG:-r06_d}; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
for lanes in 1 2 4 8; do
  timeout 600 python tools/concurrent_bench.py --lanes $lanes 16,64,128,256 600 2>&1 | grep -v "$F"
done | tee $O/callers_mrk_rank.txt
timeout 600 python tools/concurrent_bench.py --serve 16,32,64 400 2>&1 | grep -v "$F" | tee $O/callers_serve.txt
GPU_MAX_HW_QUEUES=32 timeout 600 python tools/concurrent_bench.py --serve 16,32,64 400 2>&1 | grep -v "$F" | sed 's/^/GPU_MAX_HW_QUEUES=32 /' | tee -a $O/callers_serve.txt
GPU_MAX_HW_QUEUES=32 timeout 600 python tools/concurrent_bench.py --lanes 8 64,256 600 2>&1 | grep -v "$F" | sed 's/^/GPU_MAX_HW_QUEUES=32 /' | tee -a $O/callers_mrk_rank.txt
MRK_RANK_JIT=1 timeout 900 python -m pytest -x -q -m gpu tests/test_serving_loop.py -k "native_callers" -s 2>&1 | grep "requests/s with a writer\|passed\|failed\|Error\|assert" | tee $O/pytest_stress.txt
