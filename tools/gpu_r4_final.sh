#!/bin/bash
# Round-4 closing session: the whole GPU suite, then tools/profile_r4.sh (bench lines, rocprofv3 traces, PMC passes).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|FAILED" > $O/r4_final_tests.txt
bash tools/profile_r4.sh > $O/r4_profile_session.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_smoke.txt 2>&1
