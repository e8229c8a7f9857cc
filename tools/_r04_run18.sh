timeout 900 python -m pytest tests/test_gpu_estimator.py tests/test_gpu_ops.py -k "estimator or mfdn or sfdn or wgrad or golden or stacked" -x -q 2>&1 | tail -3
python tools/inner_batch_profile.py 16 6 2>&1 | grep "batched inner step"; python tools/estimator_bench.py 2>&1 | grep backward
timeout 900 python -m pytest tests/test_gpu_edvr.py -k "inner_step or adapt or meta" -x -q 2>&1 | tail -3
