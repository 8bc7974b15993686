#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mmd or a2gnn_forward" ) > $O/r3u_tests.txt 2>&1; tail -2 $O/r3u_tests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mmd -- python tools/mmd_bench.py 40 > $O/r3u_mmd_out.txt 2> $O/r3u_mmd.err
echo "$(tail -1 $O/r3u_mmd_out.txt)"
python tools/kstats.py $O/prof_mmd k_ 2>&1 | grep -E "k_bwd|k_pairdist|k_rowstats|k_bandwidth|k_finalize" | cut -c1-140
rm -rf $O/prof_mmd
