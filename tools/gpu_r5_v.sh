#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
PYGDA_AMD_CF_DEFER_EARLY=0 PYGDA_AMD_CF_DEFER_LOGITS=0 PYGDA_AMD_SPLIT_BACKWARD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5v -- $A > $O/prof_r5v_out.txt 2> $O/prof_r5v.err
python tools/step_timeline.py $O/prof_r5v 20 2 > $O/r5v_timeline.txt 2>&1
rm -rf $O/prof_r5v/
