#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "a2gnn or cfg_a or fit or golden or dp or adam or sparse" > $O/b_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/b_tests.txt | tail -5
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
PYGDA_AMD_STAGED_SAMPLES=$1 PYGDA_AMD_SPARSE_WT_LAYOUT=$2 timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/b_bench_$1$2.json 2> $O/b_bench_$1$2.err; python -c "
import json; b=json.load(open('$O/b_bench_$1$2.json')); print('staged=$1 wt_layout=$2', round(b['ms_per_step'],4), b['config']['execution'])" || tail -3 $O/b_bench_$1$2.err; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe > $O/prof_b_out.txt 2> $O/prof_b.err
