# Round 6, measurement pass (same recipe as round 5's, tools/gpu/r04_c.sh): the whole GPU suite, the bench line of every workload,
# config 2 as written, kernel stats of the default bench command, PMC passes of c2 / c3 / c4x; then the callers table.
#   gpurun --timeout 3000 -- 'TAG=r06_g bash tools/gpu/r06_final.sh'
export TAG=${TAG:-r06_y} XGB=1
bash tools/gpu/r04_c.sh
O=gpurun_out/$TAG
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
timeout 400 python tools/concurrent_bench.py 1,4,16,32,64,128,256 600 2>&1 | grep -v "$F" | tee $O/callers_mrk_rank.txt
timeout 400 python tools/concurrent_bench.py --serve 1,4,16,32,64,128 600 2>&1 | grep -v "$F" | tee $O/callers_serve.txt
timeout 400 python tools/concurrent_bench.py --queue 1,4,16,32,64,128,256 600 2>&1 | grep -v "$F" | tee $O/callers_mrk_rank_with_queue.txt
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); $(grep throttled_usec /sys/fs/cgroup/cpu.stat 2>/dev/null)" | tee -a $O/callers_mrk_rank_with_queue.txt
find $O -name "*kernel_trace.csv" -size +1M -delete
