export DVSR_HIP_LIB=$PWD/dynavsr_amd/libdynavsr_hip_trace.so DVSR_CONV_WINO=2 DVSR_CONV_WINO3=1
timeout 120 python tools/wino_trace.py 2>&1 | grep -v amdgpu
