#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "early_cross" 2>&1 | grep -E "AssertionError|assert |^E " | head -20 | cut -c1-1500
