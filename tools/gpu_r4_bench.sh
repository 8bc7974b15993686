#!/bin/bash
# the three bench lines kept under profiles/ (after tools/profile_r4.sh's summaries are committed: the lines price
# their rooflines with those averages)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py > $O/r4_bench.json 2> $O/r4_bench.err
python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline > $O/r4_bench_powerlaw.json 2> $O/r4_bench_powerlaw.err
python bench.py --workload cfgS > $O/r4_bench_cfgS_5M.json 2> $O/r4_bench_cfgS_5M.err
