#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
timeout 1200 python -m pytest tests/test_gpu_tof.py -m gpu -x -q > gpurun_out/r02x/pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r02x/pytest.log | tail -n 4
python tools/tof_profile.py 10 2>&1 | grep TOFlow
DVSR_CONV_DMAROW=0 python tools/tof_profile.py 10 2>&1 | grep TOFlow
python tools/backbone_bench.py 2>&1 | grep -v amdgpu | head -2
