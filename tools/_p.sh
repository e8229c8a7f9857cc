#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02s
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edvr.py -m gpu -x -q -k "mdcn or dcn or backward or grad" > gpurun_out/r02s/pytest.log 2>&1
tail -n 3 gpurun_out/r02s/pytest.log
