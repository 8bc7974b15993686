#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mmd or a2gnn or grade or hipgraph or dp_ or full_size or tdss or dgsda or strurw or udagcn or two_rank" ) > gpurun_out/r3w_tests.txt 2>&1
tail -3 gpurun_out/r3w_tests.txt
for i in 1 2; do
( timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-hbm-probe --no-side-lines ) > gpurun_out/r3w_bench$i.json 2> gpurun_out/r3w_bench$i.err
python - $i <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r3w_bench{sys.argv[1]}.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])
P
done
