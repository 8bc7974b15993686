#!/bin/bash
# Round-4 measurement session on the GPU box: bench lines + rocprofv3 kernel traces + separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, LDS counters) for cfg-A (uniform and power-law stand-ins) and cfg-S (device sampler).
# Raw output under gpurun_out/, summaries (to be copied to profiles/) as gpurun_out/r4_*_rocprof_summary.*
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --profile-run"
AP="python bench.py --graph powerlaw --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --profile-run"
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
prof() {   # tag cmd...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_${tag}_out.txt 2> $O/prof_${tag}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -- "$@" > /dev/null 2> $O/pmcf_${tag}.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -- "$@" > /dev/null 2> $O/pmcw_${tag}.err
}
prof r4_cfgA $A
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcl_r4_cfgA -- $A > /dev/null 2> $O/pmcl_r4_cfgA.err
prof r4_cfgA_powerlaw $AP
prof r4_cfgS $C
python tools/summarize_rocprof.py --tag r4_cfgA --stats $O/prof_r4_cfgA --fetch $O/pmcf_r4_cfgA --write $O/pmcw_r4_cfgA --lds $O/pmcl_r4_cfgA --bench $O/prof_r4_cfgA_out.txt --cmd "$A" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r4_cfgA_powerlaw --stats $O/prof_r4_cfgA_powerlaw --fetch $O/pmcf_r4_cfgA_powerlaw --write $O/pmcw_r4_cfgA_powerlaw --bench $O/prof_r4_cfgA_powerlaw_out.txt --cmd "$AP" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r4_cfgS --stats $O/prof_r4_cfgS --fetch $O/pmcf_r4_cfgS --write $O/pmcw_r4_cfgS --bench $O/prof_r4_cfgS_out.txt --cmd "$C" --out $O > /dev/null
python tools/step_timeline.py $O/prof_r4_cfgA 20 2 > $O/r4_cfgA_timeline.txt 2>&1
rm -rf $O/prof_r4_*/ $O/pmcf_r4_*/ $O/pmcw_r4_*/ $O/pmcl_r4_*/ 2>/dev/null
ls -la $O/r4_*
timeout 300 python tools/gemm_bench.py > $O/r4_gemm_bench.jsonl 2> $O/r4_gemm_bench.err
python bench.py > $O/r4_bench.json 2> $O/r4_bench.err
python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline > $O/r4_bench_powerlaw.json 2> $O/r4_bench_powerlaw.err
python bench.py --workload cfgS > $O/r4_bench_cfgS_5M.json 2> $O/r4_bench_cfgS_5M.err
ls -la $O/r4_*
