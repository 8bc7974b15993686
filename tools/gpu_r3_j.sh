#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r3j_tests.txt 2>&1
B="python bench.py --no-side-lines --no-hbm-probe --no-cpu-baseline --steps 40 --warmup 6"
for i in 1 2; do
  timeout 200 $B > gpurun_out/r3j_bench_default_$i.json 2> gpurun_out/r3j_bench_default_$i.err
  PYGDA_AMD_TARGET_FIRST=1 timeout 200 $B > gpurun_out/r3j_bench_tfirst_$i.json 2> gpurun_out/r3j_bench_tfirst_$i.err
done
PYGDA_AMD_GRAPH_UNROLL=4 timeout 200 $B > gpurun_out/r3j_bench_unroll4.json 2> gpurun_out/r3j_bench_unroll4.err
timeout 300 python tools/cfgs_profile.py 40 > gpurun_out/r3j_cfgs_profile.txt 2>&1
tail -n 6 gpurun_out/r3j_tests.txt
for f in default_1 tfirst_1 default_2 tfirst_2 unroll4; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r3j_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],4))"; done
head -4 gpurun_out/r3j_cfgs_profile.txt
