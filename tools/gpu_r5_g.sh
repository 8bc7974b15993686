#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
for i in 1 2 3; do PYGDA_AMD_BENCH_ALLOC_DEBUG=1 $C > $O/r5g_cfgS_$i.json 2> $O/r5g_cfgS_$i.err; grep "alloc debug" $O/r5g_cfgS_$i.err | cut -c1-400; done
python - <<'PY'
import json
for f in ("r5g_cfgS_1", "r5g_cfgS_2", "r5g_cfgS_3"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in d["config"].get("host_ms_per_step_max_median")], round(d["config"].get("host_cpu_ms_per_step_median"), 3), d["config"]["hipMalloc_calls_in_timed_region"])
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r5g_tests_all.txt 2>&1
tail -6 $O/r5g_tests_all.txt
