#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for f in "--no-meta --no-inner-step" "--no-inner-step" "--no-meta" ""; do
python bench.py --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['edvr_l_bf16']; print('flags [$f]:', {k:(round(v['forward']['ms'],2), round(v['forward_backward']['ms'],2)) for k,v in e.items() if isinstance(v,dict)})"
done
