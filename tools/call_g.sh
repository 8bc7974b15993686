#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "grl_mlp or udagcn or sampler_built or mmd_golden or get_mmd or a2gnn_fit" > $O/g_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/g_tests.txt | tail -8
timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-probe > $O/g_bench.json 2> $O/g_bench.err; python -c "
import json; b=json.load(open('$O/g_bench.json')); print('cfgA', round(b['ms_per_step'],4), b.get('host_per_step'))" || tail -3 $O/g_bench.err
for f in 1 0; do PYGDA_AMD_FUSED_DOMAIN_MODEL=$f EPOCHS=60 timeout 200 python tools/other_configs_bench.py udagcn > $O/g_udagcn_$f.jsonl 2> $O/g_udagcn_$f.err; echo "fused=$f"; cat $O/g_udagcn_$f.jsonl; done
timeout 200 python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline > $O/g_cfgS.json 2> $O/g_cfgS.err; python -c "
import json; b=json.load(open('$O/g_cfgS.json')); print('cfgS', round(b['ms_per_step'],4), b['value'])" || tail -3 $O/g_cfgS.err
