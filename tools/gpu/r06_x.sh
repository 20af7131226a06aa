# r06_x: how many requests of a combined batch the one-launch kernel takes (mrk_rank's batching front), with and without op split, same box
O=gpurun_out/r06_x; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
for v in "MRK_X=1" "MRK_RANK_ONE_MAX=32" "MRK_RANK_ONE_MAX=64" "MRK_RANK_ONE_MAX=64 MRK_SPLIT_MAX_REQ=64" "MRK_RANK_ONE_MAX=128 MRK_SPLIT_MAX_REQ=32" "MRK_X=1"; do
  echo "== $v"
  env $v timeout 400 python tools/concurrent_bench.py 16,64,128,256 600 2>&1 | grep -v "$F"
done | tee $O/callers.txt
