#!/bin/bash
set -u
tag=${1:-r02e}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
run() { echo "$1: $(env $1 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt; }
run "DVSR_X=0"
run "DVSR_BWD_FORK_EVERY=1"
run "DVSR_BWD_FORK_EVERY=2"
run "DVSR_BWD_FORK_EVERY=4"
run "DVSR_BWD_FORK_EVERY=6"
run "DVSR_BWD_FORK_EVERY=10"
run "DVSR_FUSE_RES_BWD=0"
python tools/inner_bench.py 176 320 20 2>&1 | grep -v amdgpu | head -4 | tee $out/inner_bench.txt
python bench.py > $out/bench_line.json 2> $out/bench_err.txt; echo "bench rc=$?"; tail -3 $out/bench_err.txt; python -c "
import json; d=json.load(open('$out/bench_line.json'))
for k in ('value','ms_per_step'): print(k, d[k])
print('roofline', d['roofline']['frac'], d['roofline'].get('traffic_error'))
for k in ('inner_step','per_frame_pipeline','meta_step','edvr_l_bf16','cpu_baseline'): print(k, json.dumps(d.get(k))[:900])
"
