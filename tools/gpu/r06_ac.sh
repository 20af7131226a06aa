# r06_ac: is the serving queue of the committed HEAD alive on this box?  (A/B against the gang build: same sanity script)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ac; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|slow batch'
echo "== HEAD (one kernel per slot)"
MRK_LIB=$PWD/ab/head/libmrk_hip.so timeout 240 python tools/gpu/serve_sanity.py 2>&1 | grep -v "$F" | tail -30 | tee $O/sanity_head.txt
