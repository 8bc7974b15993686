#!/bin/bash
# Round-2 verification session: GPU suite, default bench line, cfg-A kernel trace + PMC passes.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/v_gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/v_gputests.txt
timeout 200 python bench.py > $O/v_bench_default.json 2> $O/v_bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/v_bench_default.json
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v_cfgA -- $A > $O/prof_v_cfgA_out.txt 2> $O/prof_v_cfgA.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_v_cfgA -- $A > /dev/null 2> $O/pmcf_v_cfgA.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_v_cfgA -- $A > /dev/null 2> $O/pmcw_v_cfgA.err
python tools/summarize_rocprof.py --tag r2b_cfgA --stats $O/prof_v_cfgA --fetch $O/pmcf_v_cfgA --write $O/pmcw_v_cfgA --bench $O/prof_v_cfgA_out.txt --cmd "$A" --out $O > /dev/null
ls -la $O/*summary*
