#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02l
timeout 600 python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > gpurun_out/r02l/op_default.txt
DVSR_CONV_TILE=0 timeout 600 python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > gpurun_out/r02l/op_th8.txt
tail -n 1 gpurun_out/r02l/op_default.txt; tail -n 1 gpurun_out/r02l/op_th8.txt
timeout 900 python -m pytest tests/test_gpu_edvr.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r02l/pytest.log 2>&1
tail -n 3 gpurun_out/r02l/pytest.log
