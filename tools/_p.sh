#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02r
timeout 600 python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > gpurun_out/r02r/op.txt
tail -n 1 gpurun_out/r02r/op.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02r/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r02r/pytest.log | tail -n 2
python bench.py --no-cpu-baseline > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02r/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['inner_step']['ms_per_step'], d['per_frame_pipeline']['ms_per_frame'])"
