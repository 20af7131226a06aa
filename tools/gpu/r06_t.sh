# r06_t: batches in flight (streams) with the round's kernels, same box
O=gpurun_out/r06_t; mkdir -p $O
for s in 2 3 4 2 3; do
  timeout 300 python bench.py --workload c2 --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>>$O/streams.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2 streams=$s', round(d['value']/1e6,1), 'M items/s')"
done | tee $O/streams.txt
for s in 2 3; do
  timeout 300 python bench.py --workload c3 --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>>$O/streams.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3 streams=$s', round(d['value']/1e6,1), 'M items/s')"
done | tee -a $O/streams.txt
