#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/r5_final_tests.txt
cat $O/r5_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
