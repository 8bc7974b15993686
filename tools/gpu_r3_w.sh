#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mmd or a2gnn or hipgraph or full_size_fit" ) > gpurun_out/r3w_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r3w_tests.txt | tail -2
for i in 1 2; do
( timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe --no-side-lines ) > gpurun_out/r3w_bench$i.json 2> gpurun_out/r3w_bench$i.err
python - $i <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r3w_bench{sys.argv[1]}.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])
P
done
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mmd -- python $GRAFT_REPO_ROOT/tools/mmd_bench.py 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py gpurun_out/prof_mmd k_ 2>&1 | grep -E "k_bwd|k_pairdist|k_rowstats|k_bandwidth|k_finalize" | cut -c1-140; rm -rf gpurun_out/prof_mmd
