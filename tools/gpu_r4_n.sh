#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for v in 1 0 1 0; do
PYGDA_AMD_FEATURES_FIRST=$v timeout 600 python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('features_first=$v ms_per_step', d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "a2gnn or hipgraph or cfg_a or minibatch" 2>&1 | grep -E "passed|failed|Error|assert" > $O/r4_n_tests.txt; cat $O/r4_n_tests.txt
