#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "graph or a2gnn or golden or capture or unroll or replay" 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "cf1_sb1 1 1" "cf1_sb0 1 0" "cf0 0 0" "cf1_sb1_b 1 1" "cf1_sb0_b 1 0" "cf0_b 0 0"; do
  set -- $v
  PYGDA_AMD_CRITICAL_FIRST=$2 PYGDA_AMD_SPLIT_BACKWARD=$3 $B > $O/r5s_$1.json 2> $O/r5s_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5s_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -5 $O/r5s_cf1_sb1.err
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5s -- $A > $O/prof_r5s_out.txt 2> $O/prof_r5s.err
python tools/step_timeline.py $O/prof_r5s 20 2 > $O/r5s_timeline.txt 2>&1
rm -rf $O/prof_r5s/
