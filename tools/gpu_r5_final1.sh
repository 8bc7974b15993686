#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/r5_final_tests.txt
cat $O/r5_final_tests.txt
bash tools/profile_r5.sh > $O/r5_profile_log.txt 2>&1
tail -12 $O/r5_profile_log.txt
