for sp in 0 1; do
echo "== DVSR_EST_SPLIT=$sp"
DVSR_EST_SPLIT=$sp python tools/estimator_bench.py 2>&1 | grep MFDN
DVSR_EST_SPLIT=$sp python tools/inner_batch_profile.py 16 6 2>&1 | grep "batched inner step"
done
DVSR_EST_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_estimator.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_ops.py -k "largest" -x -q 2>&1 | tail -3
