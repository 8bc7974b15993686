#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "grl_mlp or udagcn or sampler_built" > $O/h_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/h_tests.txt | tail -8
PYGDA_AMD_FUSED_DOMAIN_MODEL=1 EPOCHS=60 timeout 200 python tools/other_configs_bench.py udagcn > $O/h_udagcn_1.jsonl 2> $O/h_udagcn_1.err; cat $O/h_udagcn_1.jsonl; tail -2 $O/h_udagcn_1.err
timeout 200 python bench.py --workload cfgS --steps 20 --warmup 5 --no-cpu-baseline > $O/h_cfgS.json 2> $O/h_cfgS.err; python -c "
import json; b=json.load(open('$O/h_cfgS.json')); print('cfgS', round(b['ms_per_step'],4), b['value'])" || tail -3 $O/h_cfgS.err
