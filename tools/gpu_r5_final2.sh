#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > $O/r5_final_tests_full.txt 2>&1
grep -E "passed|failed|error" $O/r5_final_tests_full.txt | tail -5
