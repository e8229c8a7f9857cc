timeout 1500 python -m pytest tests/test_gpu_edvr.py -k "kink_free" -q -s 2>&1 | grep -E "kink-free|passed|failed|Error" | tail -25
