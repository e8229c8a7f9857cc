#!/bin/bash
# GPU box: parity tests + A/B of the small-grid kernels (K-split conv, ky-split wgrad) at the inner-step size.
set -u
tag=${1:-r02b}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
for cfg in "0 0" "1400 0" "0 1024" "1400 1024" "4000 1024" "700 1024"; do
  set -- $cfg
  echo "KSPLIT_BELOW=$1 KYS_BELOW=$2: $(DVSR_CONV_KSPLIT_BELOW=$1 DVSR_WGRAD_KYS_BELOW=$2 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
done
echo "NT=2 pinned: $(DVSR_CONV_KSPLIT_NT=2 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
echo "NT=1 pinned: $(DVSR_CONV_KSPLIT_NT=1 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt
python tools/op_profile.py 44 80 10 2>&1 | grep -v amdgpu > $out/per_launch_fwd44x80.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/db -o r -- python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR > $out/edvr_step_44x80.txt
python tools/rocprof_summary.py $out/db/r_results.db >> $out/edvr_step_44x80.txt
python tools/trace_dump.py $out/db/r_results.db charbonnier_partial > $out/edvr_step_44x80_timeline.txt
rm -rf $out/db
python tools/inner_bench.py 176 320 20 2>&1 | grep -v amdgpu | head -4 > $out/inner_bench.txt
cat $out/inner_bench.txt
