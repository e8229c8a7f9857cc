timeout 600 python -m pytest tests/test_gpu_ops.py -k "wgrad" -x -q 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_edvr.py -k "kink_free" -x -q -s 2>&1 | grep -E "kink-free|passed|failed|Error" | tail
for sp in 0 1; do echo "== DVSR_WGRAD_SPLIT3=$sp"; DVSR_WGRAD_SPLIT3=$sp python tools/inner_batch_profile.py 16 6 2>&1 | grep "batched inner step"; DVSR_WGRAD_SPLIT3=$sp python tools/estimator_bench.py 2>&1 | grep backward; DVSR_WGRAD_SPLIT3=$sp python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR | head -3; done
