#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x -k "gemm or skinny or tall or cfg_a_full_size_training_step or a2gnn" > $O/r5i_tests.txt 2>&1
tail -3 $O/r5i_tests.txt
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
for v in "skinny2048 2048 32768" "skinny_off 1000000000 32768" "skinny2048_wgrad8k 2048 8192" "skinny2048_b 2048 32768" "skinny_off_b 1000000000 32768" "skinny2048_wgrad8k_b 2048 8192"; do
  set -- $v
  PYGDA_AMD_SKINNY_GEMM_ROWS=$2 PYGDA_AMD_TALL_WGRAD_ROWS=$3 $B > $O/r5i_$1.json 2> $O/r5i_$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r5i_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
