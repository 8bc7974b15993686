#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
$C > $O/r5n_cfgS_1.json 2> $O/r5n_cfgS_1.err
PYGDA_AMD_BENCH_CPROFILE=$O/r5n_cprofile.txt python bench.py --workload cfgS --steps 80 --warmup 8 --no-cpu-baseline > $O/r5n_cfgS_prof.json 2> $O/r5n_cfgS_prof.err
python - <<'PY'
import json
for f in ("r5n_cfgS_1", "r5n_cfgS_prof"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(f, round(d["ms_per_step"], 3), round(c["host_cpu_ms_per_step_median"], 3), "producer cpu", [round(v, 3) for v in c["producer_cpu_ms_per_batch"]],
              "enqueue part", [round(v, 3) for v in c["producer_enqueue_cpu_ms_per_batch"]], c["host_phases"]["median_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
head -75 $O/r5n_cprofile.txt | cut -c1-160
