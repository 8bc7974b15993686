#!/bin/bash
# end-of-round validation: full GPU suite, default bench line, smoke, cfg-S profile + PMC passes on the final kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/r3f_tests.txt 2>&1
grep -E "passed|failed" $O/r3f_tests.txt | tail -1
( time timeout 900 python bench.py ) > $O/r3_bench.json 2> $O/r3_bench.err
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/r3f_smoke.txt 2>&1; tail -2 $O/r3f_smoke.txt
python bench.py --workload cfgS > $O/r3_bench_cfgS_5M.json 2> $O/r3_bench_cfgS_5M.err
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3_cfgS -- $C > $O/prof_r3_cfgS_out.txt 2> $O/prof_r3_cfgS.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_r3_cfgS -- $C > /dev/null 2> $O/pmcf_r3_cfgS.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_r3_cfgS -- $C > /dev/null 2> $O/pmcw_r3_cfgS.err
python tools/summarize_rocprof.py --tag r3_cfgS --stats $O/prof_r3_cfgS --fetch $O/pmcf_r3_cfgS --write $O/pmcw_r3_cfgS --bench $O/prof_r3_cfgS_out.txt --cmd "$C" --out $O > /dev/null
rm -rf $O/prof_r3_cfgS/ $O/pmcf_r3_cfgS/ $O/pmcw_r3_cfgS/
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["scaling_reference"]["ms_per_step"], d["scaling_reference"]["value"])
d=json.loads(open("gpurun_out/r3_bench_cfgS_5M.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
P
