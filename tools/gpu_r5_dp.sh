#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "dp or data_parallel or sampled" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload cfgS --force-dp --no-cpu-baseline > $O/r5_dp$i.json 2> $O/r5_dp$i.err
python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5_dp{sys.argv[1]}.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("force-dp", round(d["ms_per_step"], 3), c["host_ms_per_step_max_median"], {k: round(v, 3) for k, v in c["host_phases"]["median_ms"].items()})
except Exception as e:
    print("FAILED", e)
PY
done
