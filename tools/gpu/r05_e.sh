# Round 5, fifth call: which part of a k step the f32 product's matrix pipe waits for (measurement variants), and the GPU tests added since
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05_e}
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_codec.py tests/test_known_answers.py -m gpu -q -p no:cacheprovider -x > $O/pytest_some.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest_some.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/pytest_some.log | head -5
for d in 0 1 2 3; do
  MRK_ENCODER_F32_DIAG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/diag$d -o s -- python tools/encoder_bench.py --quick --precision f32 --json > $O/diag$d.log 2>&1
  f=$(find $O/diag$d -name "*kernel_stats.csv" | head -1)
  echo "diag $d: $(grep -o '"c5_batch_3840": {[^}]*}' $O/diag$d.log)"
  [ -n "$f" ] && grep "gemm_f32_mfma32" $f | awk -F, '{gsub(/"/,""); print "   ", substr($1, 1, 90), "calls", $(NF-6), "avg_ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}'
done
find $O -name "*kernel_trace.csv" -size +1M -delete
