#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=r02_z; out=gpurun_out/r02_z6; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
grep -E "passed|failed" $out/pytest.log | tail -n 2
B="--no-cpu-baseline --no-inner-step --no-split --no-meta"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/db_$c -o p -- python bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1
done
python tools/pmc_traffic.py $out/db_FETCH_SIZE/p_results.db $out/db_WRITE_SIZE/p_results.db $out/${tag}_pmc_hbm_traffic.txt $out/pmc_traffic.json
rm -rf $out/db_FETCH_SIZE $out/db_WRITE_SIZE
cp $out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $out/${tag}_bench_line.json 2> $out/bench_stderr.txt
python -c "
import json; d=json.load(open('$out/${tag}_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['inner_step']['ms_per_step'], d['per_frame_pipeline']['ms_per_frame'], d['meta_step']['ms_per_outer_iteration'], d['cpu_baseline']['value'])"
