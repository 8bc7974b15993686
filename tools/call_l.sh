#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/l_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/l_smoke.txt
timeout 150 python bench.py > $O/l_bench_default.json 2> $O/l_bench_default.err; echo "bench rc=$?"; cut -c1-330 $O/l_bench_default.json
