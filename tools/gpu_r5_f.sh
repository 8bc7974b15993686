#!/bin/bash
# Round 5, GPU session F: new tests (auto reorder, multi-seed one-pass MMD), cfg-S after the allocator pre-warm, PMC passes of
# the R-MAT aggregation (as generated / degree ordered), default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py -q -m gpu -p no:cacheprovider -s -k "auto_reorder or one_pass_mmd_is_not_worse or loader or device_sampler_equals" > $O/r5f_tests.txt 2>&1
grep -E "seed 20|passed|failed|Error" $O/r5f_tests.txt | tail -12
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
for i in 1 2 3; do $C > $O/r5f_cfgS_$i.json 2> $O/r5f_cfgS_$i.err; done
python - <<'PY'
import json
for f in ("r5f_cfgS_1", "r5f_cfgS_2", "r5f_cfgS_3"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in d["config"].get("host_ms_per_step_max_median")], round(d["config"].get("host_cpu_ms_per_step_median"), 3), d["config"]["hipMalloc_calls_in_timed_region"])
    except Exception as e:
        print(f, "FAILED", e)
PY
for m in asgen reorder; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_rmat_$m -- python tools/rmat_pmc_case.py $m > /dev/null 2> $O/pmcf_rmat_$m.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_rmat_$m -- python tools/rmat_pmc_case.py $m > /dev/null 2> $O/pmcw_rmat_$m.err
  python tools/summarize_rocprof.py --tag r5_rmat_$m --fetch $O/pmcf_rmat_$m --write $O/pmcw_rmat_$m --largest-grid --cmd "python tools/rmat_pmc_case.py $m" --out $O > /dev/null 2> $O/r5f_sum_$m.err
  rm -rf $O/pmcf_rmat_$m $O/pmcw_rmat_$m
done
ls $O/r5_rmat_* 2>/dev/null
python bench.py --no-cpu-baseline > $O/r5f_bench.json 2> $O/r5f_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5f_bench.json").read().strip().splitlines()[-1])
    print("cfg-A", d["ms_per_step"], "sustained", d["sustained"]["ms_per_step"], d["sustained"]["device_ms_per_step_p50"], "cfg-S ref", d["scaling_reference"]["ms_per_step"])
    print(json.dumps(d["roofline_hbm_regime"]["rmat_2^22"], indent=1)[:1500])
except Exception as e:
    print("FAILED", e)
PY
