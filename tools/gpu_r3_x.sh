#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tl -- $A > $O/prof_tl_out.txt 2> $O/prof_tl.err
python tools/step_timeline.py $O/prof_tl 20 2 > $O/r3x_cfgA_timeline.txt 2>&1
rm -rf $O/prof_tl
wc -l $O/r3x_cfgA_timeline.txt
