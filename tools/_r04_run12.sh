timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
bash tools/collect_profiles.sh r04_a > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log; ls gpurun_out/r04_a | wc -l
