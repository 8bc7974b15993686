#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 60 python tools/graph_double_probe.py > $O/c_double_probe.json 2> $O/c_double_probe.err; cat $O/c_double_probe.json
timeout 400 python -m pytest tests -m gpu -x -q -k "hipgraph_step or sampler_built or cfg_s or minibatch or loader" > $O/c_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/c_tests.txt | tail -5
for cfg in "0 1" "0 2" "1 2" "0 3"; do set -- $cfg
PYGDA_AMD_STAGED_SAMPLES=$1 PYGDA_AMD_GRAPH_EXECS=$2 timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/c_bench_$1$2.json 2> $O/c_bench_$1$2.err; python -c "
import json; b=json.load(open('$O/c_bench_$1$2.json')); print('staged=$1 execs=$2', round(b['ms_per_step'],4), b['config']['execution'])" || tail -3 $O/c_bench_$1$2.err; done
for c in 0 1; do PYGDA_AMD_SAMPLER_CSR=$c timeout 200 python bench.py --workload cfgS --steps 20 --warmup 5 --no-cpu-baseline > $O/c_cfgS_csr$c.json 2> $O/c_cfgS_csr$c.err; python -c "
import json; b=json.load(open('$O/c_cfgS_csr$c.json')); print('sampler_csr=$c', round(b['ms_per_step'],4), b['value'])" || tail -3 $O/c_cfgS_csr$c.err; done
