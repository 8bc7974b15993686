#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "e0l0s0 0 0 0" "e1l0s0 1 0 0" "e0l1s0 0 1 0" "e0l0s1 0 0 1" "e1l1s1 1 1 1"; do
  set -- $v
  PYGDA_AMD_CF_DEFER_EARLY=$2 PYGDA_AMD_CF_DEFER_LOGITS=$3 PYGDA_AMD_SPLIT_BACKWARD=$4 timeout 300 $B > $O/r5u_$1.json 2> $O/r5u_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5u_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4), d["config"].get("execution"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -c "capture of the training step failed" $O/r5u_$1.err
done
