#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "tall_gemm" 2>&1 | tail -3
for d in 0 2 4 6 7; do PYGDA_AMD_GEMM_DBG=$d python tools/gemm_probe.py 2>/dev/null; done
PYGDA_AMD_GEMM_SPLIT_F16=0 python tools/gemm_probe.py 2>/dev/null
