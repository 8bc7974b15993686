#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_a2gnn_fit_predict_golden" 2>&1 | grep -v "^$" | tail -70 | cut -c1-220
