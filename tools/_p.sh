#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02y2
rocprofv3 --kernel-trace --stats -d gpurun_out/r02y2/db -o r -- python tools/duf_profile.py 52 10 2>&1 | grep "DUF\|rror" | head -5
python tools/rocprof_summary.py gpurun_out/r02y2/db/r_results.db | head -22; rm -rf gpurun_out/r02y2/db
