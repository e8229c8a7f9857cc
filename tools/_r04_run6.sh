mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.txt 2>&1
tail -5 gpurun_out/gputest.txt
