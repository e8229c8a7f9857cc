#!/bin/bash
set -u
tag=${1:-r02f}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for b in 0 700 1400 2200; do
  DVSR_CONV_KSPLIT_BELOW=$b python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > $out/per_launch_fwd180x320_ksplit$b.txt
  echo "KSPLIT_BELOW=$b: $(tail -1 $out/per_launch_fwd180x320_ksplit$b.txt) rc_rb: $(grep rc_rb_a $out/per_launch_fwd180x320_ksplit$b.txt | head -1)"
done
python tools/bf16_bench.py 2>&1 | grep -v amdgpu | tee $out/bf16_modes.txt
