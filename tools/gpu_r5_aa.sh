#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5aa -- $C > $O/prof_r5aa_out.txt 2> $O/prof_r5aa.err
python tools/step_timeline.py $O/prof_r5aa 20 1 > $O/r5aa_timeline.txt 2>&1
rm -rf $O/prof_r5aa/
cut -c1-150 $O/prof_r5aa_out.txt
