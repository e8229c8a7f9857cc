#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
timeout 1500 python -m pytest tests/test_gpu_tof.py tests/test_gpu_duf.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r02x/pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r02x/pytest.log | tail -n 4
python tools/backbone_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02x/r02_z_backbones_tof_duf.txt; cat gpurun_out/r02x/r02_z_backbones_tof_duf.txt
