#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python tools/dcn_bwd_trace.py 5 44 80 2>&1 | grep -v amdgpu | grep "sampling\|lifetime\|flush"
DVSR_DCN_BWD_F32=1 python tools/dcn_bwd_trace.py 5 44 80 2>&1 | grep -v amdgpu | grep "sampling\|lifetime\|flush"
python tools/dcn_bwd_bench.py 2>&1 | grep -v amdgpu
DVSR_DCN_BWD_F32=1 python tools/dcn_bwd_bench.py 2>&1 | grep -v amdgpu
DVSR_DCN_BWD_F32=1 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mdcn" 2>&1 | tail -n 2
