#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rep() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5_q_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    c = d["config"]
    extra = c.get("host_ms_per_step_max_median") or (d.get("sustained") or {}).get("device_ms_per_step_max")
    print(sys.argv[1], round(d["ms_per_step"], 4), extra, d["host_cpu"]["intra_op_threads"], "throttled", d["host_cpu"]["cgroup_throttle_events_during_this_process"],
          "sustained", (d.get("sustained") or {}).get("ms_per_step"), (d.get("sustained") or {}).get("replays_slower_than_1p5x_median"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
C="python bench.py --workload cfgS --no-cpu-baseline"
$A > $O/r5_q_a1.json 2> $O/r5_q_a1.err; rep a1
PYGDA_AMD_CPU_THREADS=0 $A > $O/r5_q_a1_nocap.json 2> $O/r5_q_a1_nocap.err; rep a1_nocap
$A > $O/r5_q_a2.json 2> $O/r5_q_a2.err; rep a2
PYGDA_AMD_GRAPH_UNROLL=4 $A > $O/r5_q_a_u4.json 2> $O/r5_q_a_u4.err; rep a_u4
PYGDA_AMD_GRAPH_UNROLL=4 $A > $O/r5_q_a_u4b.json 2> $O/r5_q_a_u4b.err; rep a_u4b
for i in 1 2 3 4; do $C > $O/r5_q_s$i.json 2> $O/r5_q_s$i.err; rep s$i; done
PYGDA_AMD_CPU_THREADS=0 $C > $O/r5_q_s_nocap.json 2> $O/r5_q_s_nocap.err; rep s_nocap
