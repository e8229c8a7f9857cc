#!/bin/bash
# scratch: the command file of the last gpurun call (rewritten per experiment)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q
