#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02u
for w in 128 256 512 1024; do
echo "ks2_wgs $w: $(DVSR_WGRAD_KS2_WGS=$w python tools/estimator_bench.py 2>&1 | grep "forward+backward")"
done
DVSR_WGRAD_KS2_WGS=512 rocprofv3 --kernel-trace --stats -d gpurun_out/r02u/db -o r -- python tools/estimator_bench.py > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/r02u/db/r_results.db | head -14; rm -rf gpurun_out/r02u/db
