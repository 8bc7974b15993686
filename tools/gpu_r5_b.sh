#!/bin/bash
# Round 5, second GPU session: the whole GPU suite once (regressions), then cfg-S A/B (one-launch interior path on / off).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/r5b_tests_all.txt 2>&1
tail -8 $O/r5b_tests_all.txt
C="python bench.py --workload cfgS --steps 30 --warmup 8 --no-cpu-baseline"
$C > $O/r5b_cfgS_lds.json 2> $O/r5b_cfgS_lds.err
PYGDA_AMD_INTERIOR_LDS=0 $C > $O/r5b_cfgS_chain.json 2> $O/r5b_cfgS_chain.err
python - <<'PY'
import json
for f in ("r5b_cfgS_lds", "r5b_cfgS_chain"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"].get("host_ms_per_step_max_median"), d["config"].get("aggregation_launches_per_step"), {k: round(v, 4) for k, v in d["kernel_time_ms_per_step"].items() if "interior" in k or "spmm" in k})
    except Exception as e:
        print(f, "FAILED", e)
PY
P="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5b_cfgS -- $P > $O/prof_r5b_cfgS_out.txt 2> $O/prof_r5b_cfgS.err
python tools/summarize_rocprof.py --tag r5b_cfgS --stats $O/prof_r5b_cfgS --bench $O/prof_r5b_cfgS_out.txt --cmd "$P" --out $O > /dev/null 2> $O/r5b_summarize.err
rm -rf $O/prof_r5b_cfgS/
grep "k_il_\|k_spmm_range" $O/r5b_cfgS_rocprof_summary.md | cut -c1-160
