#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -q -m gpu -k "dp or data_parallel or sampled or mmd or MMD or distributed or nccl or rccl" 2>&1 | grep -E "passed|failed|error" | tail -5
