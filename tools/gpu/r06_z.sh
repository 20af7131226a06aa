# r06_z: lanes of mrk_rank's front with the one-launch kernel taking the combined batches, same box
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
O=gpurun_out/r06_z; mkdir -p $O
for l in 3 2 4 6 3; do
  timeout 300 python tools/concurrent_bench.py --lanes $l 32,64,128 600 2>&1 | grep -v "$F"
done | tee $O/lanes.txt
