mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export DVSR_CONV_WINO=2 DVSR_CONV_WINO3=1
rocprofv3 --kernel-trace --stats -d gpurun_out/db_w3 -o r -- python tools/wino_bench.py --quick > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/db_w3/r_results.db | head -12; rm -rf gpurun_out/db_w3
export DVSR_HIP_LIB=$PWD/dynavsr_amd/libdynavsr_hip_trace.so
for a in 0 2 4 8 16 6 24 30; do DVSR_CONV_ABLATE=$a timeout 120 python tools/wino_bench.py --quick 2>&1 | grep "ABLATE\|fe_rb"; done
timeout 120 python tools/wino_trace.py 2>&1 | grep -v amdgpu
