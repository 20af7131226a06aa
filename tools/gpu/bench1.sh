cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1_c.json 2> gpurun_out/bench_r1_c.log; tail -3 gpurun_out/bench_r1_c.log; cat gpurun_out/bench_r1_c.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_c -o s -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 > gpurun_out/prof_r1_c.log 2>&1
cat gpurun_out/prof_r1_c/s_kernel_stats.csv | cut -c1-200
