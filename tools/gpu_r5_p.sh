#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r5p_tests.txt
cat $O/r5p_tests.txt
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for i in 1 2; do $B > $O/r5p_cfgA_$i.json 2> $O/r5p_cfgA_$i.err; python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5p_cfgA_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print("cfgA", round(d["ms_per_step"], 4))
except Exception as e:
    print("cfgA FAILED", e)
PY
done
