#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -2
for i in $(seq 1 16); do
  PYGDA_AMD_BENCH_STALL_TRACE=15 python bench.py --workload cfgS --no-cpu-baseline > $O/r5_st2_$i.json 2> $O/r5_st2_$i.err
  python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5_st2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1], round(d["ms_per_step"], 3), [round(v, 2) for v in c["host_ms_per_step_max_median"]], c["host_phases"]["slowest_step"]["index"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
