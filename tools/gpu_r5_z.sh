#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "l1 1" "l0 0" "l1_b 1" "l0_b 0"; do
  set -- $v
  PYGDA_AMD_LOGITS_LATE=$2 $B > $O/r5z_$1.json 2> $O/r5z_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5z_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4), d["config"].get("execution")[:20])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -c "capture of the training step failed" $O/r5z_l1.err
