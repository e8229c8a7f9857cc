set -x
mkdir -p gpurun_out
export DVSR_HIP_LIB=$PWD/dynavsr_amd/libdynavsr_hip_trace.so
for a in 0 2 4 8 16 6 22 30 31; do DVSR_CONV_WINO=2 DVSR_CONV_ABLATE=$a timeout 120 python tools/wino_bench.py --quick; done > gpurun_out/wino_ablate.txt 2>&1
DVSR_CONV_WINO=2 timeout 120 python tools/wino_trace.py > gpurun_out/wino_trace.txt 2>&1
unset DVSR_HIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.txt 2>&1
tail -3 gpurun_out/gputest.txt
cat gpurun_out/wino_ablate.txt
