#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
for v in "64 4" "128 6" "128 4" "128 5" "128 8"; do set -- $v
PYGDA_AMD_MMD_BWD_TILE=$1 PYGDA_AMD_MMD_BWD_NSEG=$2 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f_$1_$2 -- python tools/mmd_bench.py 30 > $O/f_mmd_$1_$2.txt 2>&1
python tools/kstats.py $O/prof_f_$1_$2 2>/dev/null | grep -E "k_bwd|k_pairdist" | head -4 | sed "s/^/tile=$1 nseg=$2: /"; tail -1 $O/f_mmd_$1_$2.txt; done
PYGDA_AMD_MMD_BWD_TILE=128 timeout 300 python -m pytest tests -m gpu -x -q -k "mmd" > $O/f_tests.txt 2>&1; echo "tests(tile128) rc=$?"; grep -E "passed|failed|Error" $O/f_tests.txt | tail -3
timeout 300 python -m pytest tests -m gpu -x -q -k "mmd or sampler_built or cfg_s or minibatch" > $O/f_tests2.txt 2>&1; echo "tests(default) rc=$?"; grep -E "passed|failed|Error" $O/f_tests2.txt | tail -3
for c in 0 1; do PYGDA_AMD_SAMPLER_CSR=$c timeout 200 python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline > $O/f_cfgS_csr$c.json 2> $O/f_cfgS_csr$c.err; python -c "
import json; b=json.load(open('$O/f_cfgS_csr$c.json')); print('sampler_csr=$c', round(b['ms_per_step'],4), b['value'])" || tail -3 $O/f_cfgS_csr$c.err; done
