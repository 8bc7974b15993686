#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sampler.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "cfg_s or sampled or minibatch or gemm or configs3 or two_rank" > $O/r5l_tests.txt 2>&1
tail -3 $O/r5l_tests.txt
C="python bench.py --workload cfgS --steps 40 --warmup 8 --no-cpu-baseline"
for v in "split 1" "fp32 0" "split_b 1" "fp32_b 0"; do
  set -- $v
  PYGDA_AMD_GEMM_SPLIT_F16=$2 $C > $O/r5l_cfgS_$1.json 2> $O/r5l_cfgS_$1.err
done
python - <<'PY'
import json
for f in ("split", "fp32", "split_b", "fp32_b"):
    try:
        d = json.loads(open(f"gpurun_out/r5l_cfgS_{f}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(f, round(d["ms_per_step"], 3), [round(v, 3) for v in c.get("host_ms_per_step_max_median")], {k: round(v["frac"], 3) for k, v in d["roofline_dense_projection"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
