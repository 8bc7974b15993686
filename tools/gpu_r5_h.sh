#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 60 --warmup 8 --no-cpu-baseline"
for i in 1 2 3 4; do $C > $O/r5h_cfgS_$i.json 2> $O/r5h_cfgS_$i.err; done
python - <<'PY'
import json
for f in ("r5h_cfgS_1", "r5h_cfgS_2", "r5h_cfgS_3", "r5h_cfgS_4"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in c.get("host_ms_per_step_max_median")], round(c.get("host_cpu_ms_per_step_median"), 3), c["hipMalloc_calls_in_timed_region"])
        print("   ", {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in c["host_phases"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
