#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|FAILED|assert" > $O/r4_p_tests.txt; cat $O/r4_p_tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines > $O/r4_p_bench.json 2> $O/r4_p_bench.err; tail -2 $O/r4_p_bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r4_p_bench.json"))
print(d["ms_per_step"], d["value"]); print(json.dumps(d.get("roofline_mmd"))[:900])
P
