#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "graph or a2gnn or golden or capture or unroll or replay" 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "on 1" "off 0" "on_b 1" "off_b 0" "on_c 1" "off_c 0"; do
  set -- $v
  PYGDA_AMD_EARLY_CE_BACKWARD=$2 $B > $O/r5q_$1.json 2> $O/r5q_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5q_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -5 $O/r5q_on.err
