# Round 5, sixth call: workgroups per CU of the f32 128 x 128 product chosen per launch (whole rounds of resident workgroups)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_f}
mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -p no:cacheprovider -x -k "f32 or precision or c5" > $O/pytest_some.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest_some.log | tail -1)"
for r in 0 3 2 1; do
  MRK_ENCODER_F32_RESIDENT=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/res$r -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/res$r.log 2>&1
  f=$(find $O/res$r -name "*kernel_stats.csv" | head -1)
  echo "resident $r: $(grep -o '"c5_batch_3840": {[^}]*}' $O/res$r.log)"
  [ -n "$f" ] && grep "gemm_f32_mfma32" $f | awk -F, '{gsub(/"/,""); print "   ", substr($1, 1, 70), "calls", $(NF-6), "avg_ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}'
  [ "$r" = 0 ] && [ -n "$f" ] && cp $f $O/enc_f32_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -size +1M -delete
