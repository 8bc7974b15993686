#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -k "lsgan or dane" > $O/k_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert|^E " $O/k_tests.txt | tail -12
