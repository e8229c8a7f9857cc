#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_z4; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
grep -E "passed|failed" $out/pytest.log | tail -n 2
python bench.py > $out/r02_z_bench_line.json 2> $out/bench_stderr.txt
python -c "
import json; d=json.load(open('$out/r02_z_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['inner_step']['ms_per_step'], d['per_frame_pipeline']['ms_per_frame'], d['meta_step']['ms_per_outer_iteration'])"
python tools/inner_bench.py 176 320 6 2>&1 | grep -E "inner|EDVR|MFDN|full|LR|adapt_video|overlapped" > $out/r02_z_inner_step_clean.txt
python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR > $out/r02_z_edvr_step_44x80.txt
python tools/meta_bench.py 4 1 2>&1 | grep -v amdgpu > $out/r02_z_meta_train_step.txt
cat $out/r02_z_inner_step_clean.txt $out/r02_z_edvr_step_44x80.txt $out/r02_z_meta_train_step.txt
