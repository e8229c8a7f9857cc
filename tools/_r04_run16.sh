timeout 600 python -m pytest tests/test_gpu_ops.py -k "wgrad" -x -q 2>&1 | tail -4
WGRAD_BENCH_BATCHED=1 python tools/wgrad_bench.py 20 2>&1 | grep wgrad
python tools/wgrad_bench.py 20 2>&1 | grep "k3"
python tools/inner_batch_profile.py 16 6 2>&1 | grep "batched inner step"; python tools/estimator_bench.py 2>&1 | grep backward; python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR | head -2
timeout 1200 python -m pytest tests/test_gpu_edvr.py tests/test_gpu_estimator.py -k "kink_free or stacked or golden or mfdn" -x -q 2>&1 | tail -3
