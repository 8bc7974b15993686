#!/bin/bash
# Round 5, GPU session E: sampler / interior / gemm tests after the host-side and kernel edits; cfg-S twice; cfg-S kernel
# trace; the default bench line (cfg-A with the sustained figure + side lines).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "sampler or interior or gemm or skinny or tall or loader or device_batches or one_pass" > $O/r5e_tests.txt 2>&1
tail -4 $O/r5e_tests.txt
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
$C > $O/r5e_cfgS_1.json 2> $O/r5e_cfgS_1.err
$C > $O/r5e_cfgS_2.json 2> $O/r5e_cfgS_2.err
python - <<'PY'
import json
for f in ("r5e_cfgS_1", "r5e_cfgS_2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in d["config"].get("host_ms_per_step_max_median")], round(d["config"].get("host_cpu_ms_per_step_median"), 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
P="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5e_cfgS -- $P > $O/prof_r5e_cfgS_out.txt 2> $O/prof_r5e_cfgS.err
python tools/summarize_rocprof.py --tag r5e_cfgS --stats $O/prof_r5e_cfgS --bench $O/prof_r5e_cfgS_out.txt --cmd "$P" --out $O > /dev/null 2> $O/r5e_summarize.err
rm -rf $O/prof_r5e_cfgS/
grep "k_il_\|k_spmm_range\|k_skinny" $O/r5e_cfgS_rocprof_summary.md | cut -c1-150
python bench.py > $O/r5e_bench.json 2> $O/r5e_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5e_bench.json").read().strip().splitlines()[-1])
    print("cfg-A", d["ms_per_step"], d.get("sustained"), "scaling_reference", (d.get("scaling_reference") or {}).get("ms_per_step"))
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "duration_used_us", "back_to_back_launch_us", "rocprof_committed")})
except Exception as e:
    print("FAILED", e)
PY
