#!/bin/bash
# Round 5, third GPU session: the new tests, host-side profile of cfg-S, sampler stream priority, graph upload / pre-roll.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "interior or colsum or data_parallel_mmd or dp_ or rccl or two_rank or data_parallel" > $O/r5c_tests.txt 2>&1
tail -6 $O/r5c_tests.txt
timeout 300 python tools/cfgs_profile.py 30 > $O/r5c_cfgs_profile.txt 2>&1
head -4 $O/r5c_cfgs_profile.txt
C="python bench.py --workload cfgS --steps 30 --warmup 8 --no-cpu-baseline"
PYGDA_AMD_SAMPLER_PRIORITY=-1 $C > $O/r5c_cfgS_prio.json 2> $O/r5c_cfgS_prio.err
$C > $O/r5c_cfgS_default.json 2> $O/r5c_cfgS_default.err
python - <<'PY'
import json
for f in ("r5c_cfgS_prio", "r5c_cfgS_default"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"].get("host_ms_per_step_max_median"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for v in "UP1 1 0" "UP0 0 0" "UP1R8 1 8"; do
  set -- $v
  PYGDA_AMD_GRAPH_UNROLL=4 PYGDA_AMD_GRAPH_UPLOAD=$2 PYGDA_AMD_GRAPH_PREROLL=$3 timeout 200 python tools/replay_jitter.py 600 > $O/r5c_jitter_U4_$1.txt 2>&1
  echo "U4 $1"; head -5 $O/r5c_jitter_U4_$1.txt | cut -c1-220
done
PYGDA_AMD_GRAPH_UNROLL=4 python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines > $O/r5c_cfgA_U4.json 2> $O/r5c_cfgA_U4.err
python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines > $O/r5c_cfgA_U2.json 2> $O/r5c_cfgA_U2.err
python - <<'PY'
import json
for f in ("r5c_cfgA_U4", "r5c_cfgA_U2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"])
    except Exception as e:
        print(f, "FAILED", e)
PY
