#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 120 python tools/kstep_bank_bench.py > $O/a_kstep_bank.jsonl 2> $O/a_kstep_bank.err; echo "ubench rc=$?"; cat $O/a_kstep_bank.jsonl; tail -3 $O/a_kstep_bank.err
timeout 300 python -m pytest tests -m gpu -x -q -k "kstep or cfg_a or a2gnn" > $O/a_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/a_tests.txt
for b in 0 1; do PYGDA_AMD_KSTEP_BANKS=$b timeout 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/a_bench_banks$b.json 2> $O/a_bench_banks$b.err; python -c "
import json; b=json.load(open('$O/a_bench_banks$b.json')); print('banks=$b', b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac'])"; done
