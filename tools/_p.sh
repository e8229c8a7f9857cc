#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02v
timeout 1200 python -m pytest tests/test_gpu_edvr.py tests/test_gpu_ops.py tests/test_gpu_tof.py tests/test_gpu_duf.py -m gpu -x -q > gpurun_out/r02v/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r02v/pytest.log | tail -n 2
python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR
DVSR_CONV_DMA=0 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR
python tools/op_profile.py 44 80 10 2>&1 | grep -v amdgpu > gpurun_out/r02v/op44_dma.txt
DVSR_CONV_DMA=0 python tools/op_profile.py 44 80 10 2>&1 | grep -v amdgpu > gpurun_out/r02v/op44_reg.txt
tail -n 1 gpurun_out/r02v/op44_dma.txt gpurun_out/r02v/op44_reg.txt
python tools/rccl_effect.py 1 2>&1 | grep "inner step"
