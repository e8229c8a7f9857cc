#!/bin/bash
set -u
tag=${1:-r02c}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_edvr.py tests/test_gpu_estimator.py tests/test_gpu_ops.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for f in 0 1; do
  echo "BWD_FUSED=$f: $(DVSR_BWD_FUSED=$f python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
done
for w in 192 288 384 512; do
  echo "KYS_WGS=$w: $(DVSR_WGRAD_KYS_WGS=$w python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
done
echo "180x320: $(python tools/edvr_step_profile.py 180 320 5 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/db -o r -- python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR > $out/edvr_step_44x80.txt
python tools/rocprof_summary.py $out/db/r_results.db >> $out/edvr_step_44x80.txt
python tools/trace_dump.py $out/db/r_results.db charbonnier_partial > $out/edvr_step_44x80_timeline.txt
rm -rf $out/db
python tools/inner_bench.py 176 320 20 2>&1 | grep -v amdgpu | head -4 > $out/inner_bench.txt
cat $out/inner_bench.txt
