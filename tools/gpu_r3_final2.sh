#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sampler.py tests/test_gpu_fullsize.py -x -q -m gpu ) > $O/r3f2_tests.txt 2>&1
grep -E "passed|failed" $O/r3f2_tests.txt | tail -1
python bench.py > $O/r3_bench.json 2> $O/r3_bench.err
python bench.py --workload cfgS > $O/r3_bench_cfgS_5M.json 2> $O/r3_bench_cfgS_5M.err
python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline > $O/r3_bench_powerlaw.json 2> $O/r3_bench_powerlaw.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["scaling_reference"]["ms_per_step"], d["scaling_reference"]["value"])
d=json.loads(open("gpurun_out/r3_bench_cfgS_5M.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["config"]["hipMalloc_calls_in_timed_region"])
d=json.loads(open("gpurun_out/r3_bench_powerlaw.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
P
