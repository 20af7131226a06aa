# r06_i: the c2 device batch's size against whole rounds of resident workgroups (same box, same build)
O=gpurun_out/r06_i; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
for n in 3840 4096 5120 6144 7680 15360; do
  bps=$((96 * 3840 / n))
  timeout 300 python bench.py --workload c2 --requests $n --batches-per-step $bps --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>$O/req_$n.log > $O/req_$n.json
  python - <<P
import json
d = json.load(open("$O/req_$n.json"))
print("requests=$n", round(d["value"] / 1e6, 1), "M items/s", d["ms_per_step"] / $bps, "ms/batch", {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items()})
P
done | tee $O/sweep.txt
for s in 1 2 3; do
  timeout 300 python bench.py --workload c2 --requests 7680 --batches-per-step 48 --streams $s --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 0 --concurrent-callers '' 2>>$O/streams.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('requests=7680 streams=$s', round(d['value']/1e6,1), 'M items/s')"
done | tee -a $O/sweep.txt
