#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python -X faulthandler bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d.get("kernel_time_anomalies"))'
: > $O/r4_d_bench.txt
PYGDA_AMD_DEFER_WGRAD=0 timeout 120 $B 2>$O/r4_d_err_aside.txt | python -c "$pick" aside_only >> $O/r4_d_bench.txt
PYGDA_AMD_MMD_FINALIZE_ASIDE=0 timeout 120 $B 2>$O/r4_d_err_defer.txt | python -c "$pick" defer_only >> $O/r4_d_bench.txt
for i in 1 2; do
  timeout 120 $B 2>$O/r4_d_err_all.txt | python -c "$pick" all_new >> $O/r4_d_bench.txt
  PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 timeout 120 $B 2>/dev/null | python -c "$pick" bump_only >> $O/r4_d_bench.txt
  GPU_MAX_HW_QUEUES=8 timeout 120 $B 2>/dev/null | python -c "$pick" q8_all_new >> $O/r4_d_bench.txt
  GPU_MAX_HW_QUEUES=2 PYGDA_AMD_DEFER_WGRAD=0 PYGDA_AMD_MMD_FINALIZE_ASIDE=0 timeout 120 $B 2>/dev/null | python -c "$pick" q2_bump_only >> $O/r4_d_bench.txt
done
tail -25 $O/r4_d_err_aside.txt > $O/r4_d_errs.txt; echo ---- >> $O/r4_d_errs.txt; tail -25 $O/r4_d_err_defer.txt >> $O/r4_d_errs.txt
