#!/bin/bash
# Runs on the GPU box (gpurun): collects the per-round profile artefacts as TEXT under gpurun_out/
# (rocprofv3 databases are summarised in place and deleted: they exceed the 64 MiB copy-back limit).
# usage: tools/collect_profiles.sh <tag>      e.g. r01_c
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf gpurun_out/*; mkdir -p $out
python bench.py 2>&1 | tail -1 > $out/${tag}_bench_line.json
rocprofv3 --kernel-trace --stats -d $out/db1 -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inner-step --no-split > /dev/null 2>&1
python tools/rocprof_summary.py $out/db1/r_results.db > $out/${tag}_kernel_trace_bench_fwd180x320.txt; rm -rf $out/db1
rocprofv3 --kernel-trace --stats -d $out/db2 -o r -- python tools/inner_bench.py 176 320 6 2>&1 | grep -E "inner|EDVR|MFDN|full|LR" > $out/${tag}_inner_step_176x320.txt
python tools/rocprof_summary.py $out/db2/r_results.db >> $out/${tag}_inner_step_176x320.txt; rm -rf $out/db2
python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > $out/${tag}_per_launch_fwd180x320.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/db_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inner-step --no-split > /dev/null 2>&1
done
python tools/pmc_traffic.py $out/db_FETCH_SIZE/p_results.db $out/db_WRITE_SIZE/p_results.db $out/${tag}_pmc_hbm_traffic.txt $out/pmc_traffic.json
rm -rf $out/db_FETCH_SIZE $out/db_WRITE_SIZE
python tools/metrics_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_metrics_720x1280.txt
rocprofv3 --kernel-trace --stats -d $out/db3 -o m -- python tools/metrics_bench.py > /dev/null 2>&1
python tools/rocprof_summary.py $out/db3/m_results.db | head -8 >> $out/${tag}_metrics_720x1280.txt; rm -rf $out/db3
python tools/degrade_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_degradation.txt
python tools/meta_bench.py 4 1 2>&1 | grep -v amdgpu > $out/${tag}_meta_train_step.txt
du -sh gpurun_out
