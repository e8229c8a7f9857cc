#!/bin/bash
set -u
tag=${1:-r02d}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
run() { echo "$1: $(env $1 python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR)" | tee -a $out/ab_44x80.txt; }
run "DVSR_X=0"
run "DVSR_WGRAD_KYS_WGS=192"
run "DVSR_WGRAD_KYS_WGS=384"
run "DVSR_WGRAD_KYS_WGS=512"
run "DVSR_WGRAD_KYS_WGS=768"
run "DVSR_WGRAD_NOFLUSH=1"
run "DVSR_WGRAD_NOFLUSH=1 DVSR_BWD_FUSED=1"
run "DVSR_BWD_STREAMS=0"
run "DVSR_WGRAD_KYS_BELOW=0"
python tools/estimator_bench.py 2>&1 | grep -v amdgpu | tee -a $out/ab_44x80.txt
DVSR_WGRAD_NOFLUSH=1 python tools/estimator_bench.py 2>&1 | grep -v amdgpu | sed 's/^/NOFLUSH /' | tee -a $out/ab_44x80.txt
