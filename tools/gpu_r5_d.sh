#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
$C > $O/r5d_default.json 2> $O/r5d_default.err
PYGDA_AMD_MMD_PREFETCH=0 $C > $O/r5d_noprefetch.json 2> $O/r5d_noprefetch.err
PYGDA_AMD_INTERIOR_LDS=0 $C > $O/r5d_chain.json 2> $O/r5d_chain.err
PYGDA_AMD_SWITCH_US=200 $C > $O/r5d_switch200.json 2> $O/r5d_switch200.err
python - <<'PY'
import json
for f in ("r5d_default", "r5d_noprefetch", "r5d_chain", "r5d_switch200"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in d["config"].get("host_ms_per_step_max_median")], round(d["config"].get("host_cpu_ms_per_step_median"), 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
