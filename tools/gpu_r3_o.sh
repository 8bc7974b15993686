#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r3o_tests.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r3o_smoke.txt 2>&1
grep -E "passed|failed" gpurun_out/r3o_tests.txt; tail -4 gpurun_out/r3o_bench.err; tail -3 gpurun_out/r3o_smoke.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3o_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["scaling_reference"]["ms_per_step"], d["scaling_reference"]["value"])
print(json.dumps(d["roofline_hbm_regime"])[:900])
P
