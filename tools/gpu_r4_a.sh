#!/bin/bash
# Round-4 GPU session A: parity of the changed kernels (MMD upper-triangular weights + transposed staging, step-counter
# bump, finalize on a side stream), then the cfg-A line with each change switched off in turn.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "mmd or a2gnn or full_size or adagcn or grade or dp_ or captured" 2>&1 | tail -15 > $O/r4_a_tests.txt
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d.get("kernel_time_anomalies"), json.dumps(d["roofline"])[:300])'
for i in 1 2; do $B 2>$O/r4_a_b0.err | python -c "$pick" default >> $O/r4_a_bench.txt; done
PYGDA_AMD_MMD_FULL_G=1 $B 2>/dev/null | python -c "$pick" full_G >> $O/r4_a_bench.txt
PYGDA_AMD_BUMP_AT_START=0 $B 2>/dev/null | python -c "$pick" no_bump >> $O/r4_a_bench.txt
PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" no_aside >> $O/r4_a_bench.txt
PYGDA_AMD_MMD_FULL_G=1 PYGDA_AMD_BUMP_AT_START=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" round3_equiv >> $O/r4_a_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mmd -- python tools/mmd_bench.py 40 > $O/r4_a_mmd_upper.txt 2>&1
python tools/kstats.py $O/prof_mmd >> $O/r4_a_mmd_upper.txt 2>&1
PYGDA_AMD_MMD_FULL_G=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mmd_full -- python tools/mmd_bench.py 40 > $O/r4_a_mmd_full.txt 2>&1
python tools/kstats.py $O/prof_mmd_full >> $O/r4_a_mmd_full.txt 2>&1
rm -rf $O/prof_mmd $O/prof_mmd_full
