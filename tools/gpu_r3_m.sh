#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --no-side-lines --no-hbm-probe --no-cpu-baseline --steps 40 --warmup 6"
for i in 1 2; do
  timeout 200 $B > gpurun_out/r3m_bench_default_$i.json 2> gpurun_out/r3m_bench_default_$i.err
  PYGDA_AMD_TARGET_L0_FIRST=1 timeout 200 $B > gpurun_out/r3m_bench_l0first_$i.json 2> gpurun_out/r3m_bench_l0first_$i.err
done
PYGDA_AMD_TARGET_L0_FIRST=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "a2gnn_fit or fit_captured" > gpurun_out/r3m_tests.txt 2>&1
for f in default_1 l0first_1 default_2 l0first_2; do python -c "
import json
d=json.loads(open('gpurun_out/r3m_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],4))"; done
tail -3 gpurun_out/r3m_tests.txt
