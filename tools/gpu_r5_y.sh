#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "early_cross or hipgraph_step or fit_predict_golden" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "d1 1" "d0 0" "d1_b 1" "d0_b 0" "d1_c 1" "d0_c 0"; do
  set -- $v
  PYGDA_AMD_DEFER_EARLY=$2 $B > $O/r5y_$1.json 2> $O/r5y_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5y_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4), d["config"].get("execution")[:20])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
