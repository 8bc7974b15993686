#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
( timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe --no-side-lines ) > gpurun_out/r3w_bench$i.json 2> gpurun_out/r3w_bench$i.err
python - $i <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r3w_bench{sys.argv[1]}.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])
P
done
