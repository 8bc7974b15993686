#!/bin/bash
# Round-5 measurement session on the GPU box: bench lines + rocprofv3 kernel traces + separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, LDS counters) for cfg-A (uniform and power-law stand-ins) and cfg-S (device sampler, one-launch interior K-step).
# Raw output under gpurun_out/, summaries (to be copied to profiles/) as gpurun_out/r5_*_rocprof_summary.*
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
AP="python bench.py --graph powerlaw --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
prof() {   # tag cmd...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_${tag}_out.txt 2> $O/prof_${tag}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -- "$@" > /dev/null 2> $O/pmcf_${tag}.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -- "$@" > /dev/null 2> $O/pmcw_${tag}.err
}
prof r5_cfgA $A
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcl_r5_cfgA -- $A > /dev/null 2> $O/pmcl_r5_cfgA.err
prof r5_cfgA_powerlaw $AP
prof r5_cfgS $C
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcl_r5_cfgS -- $C > /dev/null 2> $O/pmcl_r5_cfgS.err
python tools/summarize_rocprof.py --tag r5_cfgA --stats $O/prof_r5_cfgA --fetch $O/pmcf_r5_cfgA --write $O/pmcw_r5_cfgA --lds $O/pmcl_r5_cfgA --bench $O/prof_r5_cfgA_out.txt --cmd "$A" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r5_cfgA_powerlaw --stats $O/prof_r5_cfgA_powerlaw --fetch $O/pmcf_r5_cfgA_powerlaw --write $O/pmcw_r5_cfgA_powerlaw --bench $O/prof_r5_cfgA_powerlaw_out.txt --cmd "$AP" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r5_cfgS --stats $O/prof_r5_cfgS --fetch $O/pmcf_r5_cfgS --write $O/pmcw_r5_cfgS --lds $O/pmcl_r5_cfgS --lds-kernel k_il_lds --bench $O/prof_r5_cfgS_out.txt --cmd "$C" --out $O > /dev/null
python tools/step_timeline.py $O/prof_r5_cfgA 20 2 > $O/r5_cfgA_timeline.txt 2>&1
rm -rf $O/prof_r5_*/ $O/pmcf_r5_*/ $O/pmcw_r5_*/ $O/pmcl_r5_*/ 2>/dev/null
ls -la $O/r5_*
