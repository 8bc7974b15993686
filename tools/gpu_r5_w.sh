#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "graph or a2gnn or golden or capture or unroll or replay or early" 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for i in 1 2; do $B > $O/r5w_$i.json 2> $O/r5w_$i.err; python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5w_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print("cfgA", round(d["ms_per_step"], 4))
except Exception as e:
    print("cfgA FAILED", e)
PY
done
