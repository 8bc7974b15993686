#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  extra="--no-cpu-baseline"; if [ $i -gt 8 ]; then extra=""; fi
  PYGDA_AMD_BENCH_STALL_TRACE=15 python bench.py --workload cfgS $extra > $O/r5_stall_$i.json 2> $O/r5_stall_$i.err
  python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5_stall_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1], round(d["ms_per_step"], 3), [round(v, 2) for v in c["host_ms_per_step_max_median"]], c["host_phases"]["slowest_step"]["index"], grep if False else "")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -c "Timeout" $O/r5_stall_$i.err
done
