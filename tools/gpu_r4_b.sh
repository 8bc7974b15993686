#!/bin/bash
# Round-4 GPU session B: deferred weight gradients / step-counter bump / MMD finalize aside, A/B with repeats; timeline.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "a2gnn or full_size_training or full_size_fit or captured or tdss or dgsda" 2>&1 | tail -8 > $O/r4_b_tests.txt
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d.get("kernel_time_anomalies"))'
: > $O/r4_b_bench.txt
for i in 1 2 3; do
  $B 2>$O/r4_b_err.txt | python -c "$pick" all_new >> $O/r4_b_bench.txt
  PYGDA_AMD_DEFER_WGRAD=0 $B 2>/dev/null | python -c "$pick" no_defer >> $O/r4_b_bench.txt
done
for i in 1 2; do
  PYGDA_AMD_BUMP_AT_START=0 $B 2>/dev/null | python -c "$pick" no_bump >> $O/r4_b_bench.txt
  PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" no_aside >> $O/r4_b_bench.txt
  PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_BUMP_AT_START=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" all_off >> $O/r4_b_bench.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines > $O/r4_b_prof_out.txt 2> $O/r4_b_prof.err
python tools/step_timeline.py $O/prof_b 20 2 > $O/r4_b_timeline.txt 2>&1
rm -rf $O/prof_b
