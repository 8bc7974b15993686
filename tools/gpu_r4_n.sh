#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mmd" 2>&1 | grep -E "passed|failed|Error|assert" > $O/r4_n_tests.txt; cat $O/r4_n_tests.txt
rm -rf $O/mmdk; timeout 300 python tools/mmd_kernels.py $O/mmdk > $O/r4_n_kernels.json 2>&1; cat $O/r4_n_kernels.json; rm -rf $O/mmdk
timeout 600 python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
