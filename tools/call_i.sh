#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "hipgraph_step or a2gnn_fit or grl_mlp_ce_dropout" > $O/i_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/i_tests.txt | tail -8
for u in 1 2 4 8; do PYGDA_AMD_GRAPH_UNROLL=$u timeout 120 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-hbm-probe > $O/i_bench_$u.json 2> $O/i_bench_$u.err; python -c "
import json; b=json.load(open('$O/i_bench_$u.json')); print('unroll=$u', round(b['ms_per_step'],4), b.get('host_per_step'))" || tail -3 $O/i_bench_$u.err; done
