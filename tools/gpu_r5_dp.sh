#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py --workload cfgS --force-dp --no-cpu-baseline > $O/r5_dp1.json 2> $O/r5_dp1.err; tail -c 300 $O/r5_dp1.err
timeout 600 python bench.py --gpus 2 --share-gpus --steps 10 --warmup 3 --no-cpu-baseline > $O/r5_dp2.json 2> $O/r5_dp2.err; tail -c 600 $O/r5_dp2.err
python - <<'PY'
import json
for f in ("r5_dp1", "r5_dp2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["n_gpus"], round(d["ms_per_step"], 3), d["config"].get("parallelism"), d.get("rccl_ranks_seen"), d.get("collectives"), d.get("functional_check", "")[:60])
    except Exception as e:
        print(f, "FAILED", e)
PY
