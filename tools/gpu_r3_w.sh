#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "sparse or a2gnn or hipgraph or full_size or kstep_lds_cfg_a" ) > gpurun_out/r3w_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r3w_tests.txt | tail -2
for v in 1 0 1 0; do
( PYGDA_AMD_SPARSE_COLMAJOR=$v timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe --no-side-lines ) > gpurun_out/r3w_bench$v.json 2> gpurun_out/r3w_bench$v.err
python - $v <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r3w_bench{sys.argv[1]}.json").read().strip().splitlines()[-1]); print("colmajor", sys.argv[1], d["ms_per_step"], d["value"])
P
done
