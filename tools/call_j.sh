#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/j_gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|FAILED" $O/j_gputests.txt | tail -8
