#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
for i in 1 2; do $C > $O/r5m_cfgS_$i.json 2> $O/r5m_cfgS_$i.err; done
python - <<'PY'
import json
for f in ("r5m_cfgS_1", "r5m_cfgS_2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in c.get("host_ms_per_step_max_median")], round(c["host_cpu_ms_per_step_median"], 3), "producer cpu ms/batch", [round(v, 3) for v in c["producer_cpu_ms_per_batch"]])
    except Exception as e:
        print(f, "FAILED", e)
PY
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "w8192 8192" "w4096 4096" "w32768 32768" "w8192_b 8192" "w4096_b 4096" "w32768_b 32768"; do
  set -- $v
  PYGDA_AMD_TALL_WGRAD_ROWS=$2 $B > $O/r5m_$1.json 2> $O/r5m_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5m_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
