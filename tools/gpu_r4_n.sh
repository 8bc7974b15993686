#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|FAILED" > $O/r4_n_tests.txt; cat $O/r4_n_tests.txt
for w in 384 640; do
  rm -rf $O/mmdk; PYGDA_AMD_MMD_FUSED_WGS=$w timeout 300 python tools/mmd_kernels.py $O/mmdk 30 | cut -c1-400; rm -rf $O/mmdk
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
