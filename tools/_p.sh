#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for f in 1 2 3 4 6; do
echo "fork_every $f: $(DVSR_BWD_FORK_EVERY=$f python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR) | $(DVSR_BWD_FORK_EVERY=$f python tools/rccl_effect.py 1 2>&1 | grep 'inner step')"
done
