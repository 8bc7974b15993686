#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "mmd or dp or data_parallel or a2gnn_forward" ) > $O/r3q_tests.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mmd -- python tools/mmd_bench.py 50 > $O/r3q_mmd_out.txt 2> $O/r3q_mmd.err
python tools/kstats.py $O/prof_mmd k_ > $O/r3q_mmd_kstats.txt 2>&1
rm -rf $O/prof_mmd
grep -E "passed|failed" $O/r3q_tests.txt; cat $O/r3q_mmd_kstats.txt | cut -c1-150
