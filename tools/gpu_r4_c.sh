#!/bin/bash
# Round-4 GPU session C: weight gradients / MMD finalize on the BORROWED statistics stream (no fifth stream), and the
# number of hardware queues the HIP runtime multiplexes streams onto (GPU_MAX_HW_QUEUES, default 4).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "a2gnn_fit or full_size_training or full_size_fit or captured" 2>&1 | tail -4 > $O/r4_c_tests.txt
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d.get("kernel_time_anomalies"))'
: > $O/r4_c_bench.txt
for i in 1 2; do
  $B 2>$O/r4_c_err.txt | python -c "$pick" all_new >> $O/r4_c_bench.txt
  PYGDA_AMD_DEFER_WGRAD=0 $B 2>/dev/null | python -c "$pick" no_defer >> $O/r4_c_bench.txt
  PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" no_defer_no_aside >> $O/r4_c_bench.txt
  GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | python -c "$pick" q8_all_new >> $O/r4_c_bench.txt
  GPU_MAX_HW_QUEUES=8 PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 PYGDA_AMD_BUMP_AT_START=0 $B 2>/dev/null | python -c "$pick" q8_all_off >> $O/r4_c_bench.txt
  GPU_MAX_HW_QUEUES=2 PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 $B 2>/dev/null | python -c "$pick" q2_no_defer >> $O/r4_c_bench.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines > $O/r4_c_prof_out.txt 2> $O/r4_c_prof.err
python tools/step_timeline.py $O/prof_c 20 2 > $O/r4_c_timeline.txt 2>&1
rm -rf $O/prof_c
