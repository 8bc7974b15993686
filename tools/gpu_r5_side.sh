#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-sustained"
for v in "ring1 1" "alloc1 0" "ring2 1" "alloc2 0" "ring3 1"; do
  set -- $v
  PYGDA_AMD_BENCH_STALL_TRACE=15 PYGDA_AMD_LOADER_RECYCLE=$2 $B > $O/r5_side_$1.json 2> $O/r5_side_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5_side_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    s = d["scaling_reference"]
    print(sys.argv[1], "cfgA", round(d["ms_per_step"], 4), "cfgS", round(s["ms_per_step"], 3), s["host_ms_per_step_max_median"], s["host_slowest_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -c "Timeout" $O/r5_side_$1.err
done
