#!/bin/bash
# Round 5, first GPU session: the new tests (one-launch interior K-step, SAGE/GIN/GAT goldens, ADVICE fixes), then
# cfg-S with and without the one-launch path, and a kernel trace of cfg-S.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampler.py -q -m gpu -k "interior" -x > $O/r5a_tests_interior.txt 2>&1
tail -5 $O/r5a_tests_interior.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sage_gin_gat or adam or mmd or rccl or direct" > $O/r5a_tests_parity.txt 2>&1
tail -5 $O/r5a_tests_parity.txt
C="python bench.py --workload cfgS --steps 30 --warmup 8 --no-cpu-baseline"
$C > $O/r5a_cfgS_lds.json 2> $O/r5a_cfgS_lds.err
PYGDA_AMD_INTERIOR_LDS=0 $C > $O/r5a_cfgS_chain.json 2> $O/r5a_cfgS_chain.err
$C > $O/r5a_cfgS_lds2.json 2> $O/r5a_cfgS_lds2.err
python - <<'PY'
import json
for f in ("r5a_cfgS_lds", "r5a_cfgS_chain", "r5a_cfgS_lds2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"].get("host_ms_per_step_max_median"), d["config"].get("aggregation_launches_per_step"), d["config"].get("aggregation_paths"))
    except Exception as e:
        print(f, "FAILED", e)
PY
P="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5a_cfgS -- $P > $O/prof_r5a_cfgS_out.txt 2> $O/prof_r5a_cfgS.err
python tools/summarize_rocprof.py --tag r5a_cfgS --stats $O/prof_r5a_cfgS --bench $O/prof_r5a_cfgS_out.txt --cmd "$P" --out $O > /dev/null 2> $O/r5a_summarize.err
rm -rf $O/prof_r5a_cfgS/
head -40 $O/r5a_cfgS_rocprof_summary.md
