#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3b_cfgS -- $C > $O/prof_r3b_cfgS_out.txt 2> $O/prof_r3b_cfgS.err
python tools/summarize_rocprof.py --tag r3b_cfgS --stats $O/prof_r3b_cfgS --bench $O/prof_r3b_cfgS_out.txt --cmd "$C" --out $O > /dev/null
rm -rf $O/prof_r3b_cfgS
timeout 300 python tools/gemm_bench.py > $O/r3_gemm_bench.jsonl 2> $O/r3_gemm_bench.err
$C > $O/r3k_cfgS_bench.json 2> $O/r3k_cfgS_bench.err
head -48 $O/r3b_cfgS_rocprof_summary.md | cut -c1-170
head -c 400 $O/r3k_cfgS_bench.json
