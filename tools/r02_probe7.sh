#!/bin/bash
set -u
tag=${1:-r02g}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/db -o r -- python tools/estimator_bench.py 176 320 10 > $out/est.txt 2>&1
python tools/rocprof_summary.py $out/db/r_results.db > $out/estimator_kernels.txt
python tools/trace_dump.py $out/db/r_results.db rowmean_kernel > $out/estimator_timeline.txt
rm -rf $out/db
grep MFDN $out/est.txt
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
t0 = time.time(); r = bench.edvr_l_rates(dev); print({k: (v['forward']['ms'], v['forward_backward']['ms']) for k, v in r.items() if isinstance(v, dict)}, time.time() - t0)
PY
