#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -8
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
for i in 1 2; do
  $C > $O/r5o_ring_$i.json 2> $O/r5o_ring_$i.err
  PYGDA_AMD_LOADER_RECYCLE=0 $C > $O/r5o_alloc_$i.json 2> $O/r5o_alloc_$i.err
done
python - <<'PY'
import json
for f in ("r5o_ring_1", "r5o_alloc_1", "r5o_ring_2", "r5o_alloc_2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(f, round(d["ms_per_step"], 3), "host", [round(v, 3) for v in c.get("host_ms_per_step_max_median")], "cpu", round(c["host_cpu_ms_per_step_median"], 3), "producer cpu", [round(v, 3) for v in c["producer_cpu_ms_per_batch"]],
              "enqueue part", [round(v, 3) for v in c["producer_enqueue_cpu_ms_per_batch"]], {k: round(v, 3) for k, v in c["host_phases"]["median_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $O/r5o_ring_1.err
