#!/bin/bash
# GPU box: gpu tests, then the launch-level view of the inner-step-sized EDVR forward+backward (44x80).
set -u
tag=${1:-r02a}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python tools/op_profile.py 44 80 10 2>&1 | grep -v amdgpu > $out/per_launch_fwd44x80.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/db -o r -- python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR > $out/edvr_step_44x80.txt
python tools/rocprof_summary.py $out/db/r_results.db >> $out/edvr_step_44x80.txt
python tools/trace_dump.py $out/db/r_results.db charbonnier_partial > $out/edvr_step_44x80_timeline.txt
python tools/stream_timeline.py $out/db/r_results.db 0.5 > $out/edvr_step_44x80_streams.txt
rm -rf $out/db
python tools/inner_bench.py 176 320 20 2>&1 | grep -v amdgpu > $out/inner_bench.txt
python tools/estimator_bench.py 2>&1 | grep -v amdgpu > $out/estimator_bench.txt
cat $out/edvr_step_44x80.txt | head -5; cat $out/inner_bench.txt
