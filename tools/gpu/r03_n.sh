# round 3, pass n: phase clocks of the fused assembly kernel - a loaded batch and 32 requests (each workgroup alone on a CU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_n
mkdir -p $O
MRK_DEFINES=MRK_PHASE_CLOCKS python -c "from metarank_amd import _native; _native.build(force=True)" > $O/build.log 2>&1; tail -2 $O/build.log
export MRK_RANK_JIT=1 MRK_FUSED_SPLIT=1
timeout 600 python tools/phase_clocks.py c2 32 > $O/phase_c2_32.txt 2>&1; cat $O/phase_c2_32.txt
timeout 600 python tools/phase_clocks.py c2 > $O/phase_c2_full.txt 2>&1; cat $O/phase_c2_full.txt
