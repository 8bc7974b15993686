#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_configs.py -x -q -m gpu -k "sampler or loader or sampled or cfg_s or minibatch or two_rank" ) > gpurun_out/r3z_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r3z_tests.txt | tail -2
for h in 1 0 1 0; do
PYGDA_AMD_GATHER_AHEAD=$h python bench.py --workload cfgS --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead $h', d['ms_per_step'], d['value'], d['config']['host_ms_per_step_max_median'])"
done
