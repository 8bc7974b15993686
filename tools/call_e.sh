#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 60 python tools/graph_host_cost_probe.py > $O/e_host_probe.json 2> $O/e_host_probe.err; cat $O/e_host_probe.json; tail -2 $O/e_host_probe.err
PYGDA_AMD_ASYNC_LAUNCH=0 timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/e_bench_one.json 2> $O/e_bench_one.err; python -c "
import json; b=json.load(open('$O/e_bench_one.json')); print('one graph', round(b['ms_per_step'],4), b.get('host_per_step'))" || tail -3 $O/e_bench_one.err
PYGDA_AMD_ASYNC_LAUNCH=0 PYGDA_AMD_SPLIT_GRAPHS=1 timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/e_bench_split.json 2> $O/e_bench_split.err; python -c "
import json; b=json.load(open('$O/e_bench_split.json')); print('split', round(b['ms_per_step'],4), b.get('host_per_step'))" || tail -3 $O/e_bench_split.err
