set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -k winograd -x -q 2>&1 | tail -15
DVSR_CONV_WINO=2 DVSR_CONV_WINO3=1 timeout 300 python tools/wino_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/wino3_bench.txt
DVSR_CONV_WINO=2 DVSR_CONV_WINO3=0 timeout 300 python tools/wino_bench.py --quick 2>&1 | grep -v amdgpu | tee -a gpurun_out/wino3_bench.txt
