#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d.get("kernel_time_anomalies"))'
: > $O/r4_e_bench.txt
for i in 1 2 3; do
  timeout 120 $B 2>$O/r4_e_err.txt | python -c "$pick" default >> $O/r4_e_bench.txt
  PYGDA_AMD_LOGITS_ON_SRC=1 timeout 120 $B 2>/dev/null | python -c "$pick" logits_on_src >> $O/r4_e_bench.txt
  PYGDA_AMD_GRAPH_UNROLL=1 timeout 120 $B 2>/dev/null | python -c "$pick" unroll1 >> $O/r4_e_bench.txt
  PYGDA_AMD_GRAPH_UNROLL=3 timeout 120 $B 2>/dev/null | python -c "$pick" unroll3 >> $O/r4_e_bench.txt
done
PYTORCH_HIP_ALLOC_CONF=roundup_power2_divisions:8 timeout 300 python bench.py --workload cfgS --no-cpu-baseline 2>$O/r4_e_cfgs_err.txt | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("cfgS_roundup", d["ms_per_step"], d["config"]["hipMalloc_calls_in_timed_region"], d["config"]["host_ms_per_step_max_median"])' >> $O/r4_e_bench.txt
timeout 300 python bench.py --workload cfgS --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("cfgS_default", d["ms_per_step"], d["config"]["hipMalloc_calls_in_timed_region"], d["config"]["host_ms_per_step_max_median"])' >> $O/r4_e_bench.txt
python -m pytest tests -m gpu -q -x -k "a2gnn_fit or full_size_training or captured or mmd" 2>&1 | tail -3 > $O/r4_e_tests.txt
