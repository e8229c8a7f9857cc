cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/db16 -o r -- python tools/inner_batch_profile.py 16 4 > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/db16/r_results.db | head -34; rm -rf gpurun_out/db16
