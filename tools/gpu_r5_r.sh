#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5r -- $A > $O/prof_r5r_out.txt 2> $O/prof_r5r.err
python tools/step_timeline.py $O/prof_r5r 20 2 > $O/r5r_timeline.txt 2>&1
rm -rf $O/prof_r5r/
cat $O/prof_r5r_out.txt | cut -c1-200
