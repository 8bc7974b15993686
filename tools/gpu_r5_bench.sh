#!/bin/bash
# the bench lines kept under profiles/ (after tools/profile_r5.sh's summaries are committed: the lines quote their
# averages as hash-checked side figures)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py > $O/r5_bench.json 2> $O/r5_bench.err
python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline > $O/r5_bench_powerlaw.json 2> $O/r5_bench_powerlaw.err
python bench.py --workload cfgS > $O/r5_bench_cfgS_5M.json 2> $O/r5_bench_cfgS_5M.err
