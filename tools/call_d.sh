#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q -k "fit or golden or hipgraph or captured or sampler_built or cfg_s or minibatch or loader" > $O/d_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/d_tests.txt | tail -5
for a in 0 1 0 1; do
PYGDA_AMD_ASYNC_LAUNCH=$a timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/d_bench_$a.json 2> $O/d_bench_$a.err; python -c "
import json; b=json.load(open('$O/d_bench_$a.json')); print('async=$a', round(b['ms_per_step'],4), b['config']['execution'])" || tail -3 $O/d_bench_$a.err; done
for c in 0 1; do PYGDA_AMD_SAMPLER_CSR=$c timeout 200 python bench.py --workload cfgS --steps 20 --warmup 5 --no-cpu-baseline > $O/d_cfgS_csr$c.json 2> $O/d_cfgS_csr$c.err; python -c "
import json; b=json.load(open('$O/d_cfgS_csr$c.json')); print('sampler_csr=$c', round(b['ms_per_step'],4), b['value'])" || tail -3 $O/d_cfgS_csr$c.err; done
