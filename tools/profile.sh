#!/bin/bash
# Measurement session on the GPU box: bench lines + rocprofv3 kernel traces + separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# LDS counters) for cfg-A (uniform and power-law stand-ins) and cfg-S (device sampler, one-launch interior K-step), and the
# HBM-regime aggregations (uniform 5 M nodes, R-MAT 2^22 as generated / degree ordered).
#   usage: bash tools/profile.sh <round tag, e.g. r6> [cfgA] [powerlaw] [cfgS] [hbm]     (no selection = all)
# Raw output under gpurun_out/; summaries (to be copied to profiles/) as gpurun_out/<tag>_*_rocprof_summary.*
# Every summary carries `library_sha16` of the libgda_hip.so it was made with (tools/summarize_rocprof.py).
set -u
R=${1:?round tag}; shift
WHAT=${*:-cfgA powerlaw cfgS hbm}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
AP="python bench.py --graph powerlaw --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained --profile-run"
C="python bench.py --workload cfgS --steps 30 --warmup 5 --no-cpu-baseline --profile-run"
LDS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
prof() {   # tag cmd...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_${tag}_out.txt 2> $O/prof_${tag}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -- "$@" > /dev/null 2> $O/pmcf_${tag}.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -- "$@" > /dev/null 2> $O/pmcw_${tag}.err
}
for w in $WHAT; do
  case $w in
    cfgA)
      prof ${R}_cfgA $A
      rocprofv3 --pmc $LDS --output-format csv -d $O/pmcl_${R}_cfgA -- $A > /dev/null 2> $O/pmcl_${R}_cfgA.err
      python tools/summarize_rocprof.py --tag ${R}_cfgA --stats $O/prof_${R}_cfgA --fetch $O/pmcf_${R}_cfgA --write $O/pmcw_${R}_cfgA --lds $O/pmcl_${R}_cfgA --bench $O/prof_${R}_cfgA_out.txt --cmd "$A" --out $O > /dev/null
      python tools/step_timeline.py $O/prof_${R}_cfgA 20 2 > $O/${R}_cfgA_timeline.txt 2>&1 ;;
    powerlaw)
      prof ${R}_cfgA_powerlaw $AP
      python tools/summarize_rocprof.py --tag ${R}_cfgA_powerlaw --stats $O/prof_${R}_cfgA_powerlaw --fetch $O/pmcf_${R}_cfgA_powerlaw --write $O/pmcw_${R}_cfgA_powerlaw --bench $O/prof_${R}_cfgA_powerlaw_out.txt --cmd "$AP" --out $O > /dev/null ;;
    cfgS)
      prof ${R}_cfgS $C
      rocprofv3 --pmc $LDS --output-format csv -d $O/pmcl_${R}_cfgS -- $C > /dev/null 2> $O/pmcl_${R}_cfgS.err
      python tools/summarize_rocprof.py --tag ${R}_cfgS --stats $O/prof_${R}_cfgS --fetch $O/pmcf_${R}_cfgS --write $O/pmcw_${R}_cfgS --lds $O/pmcl_${R}_cfgS --lds-kernel k_il_lds --bench $O/prof_${R}_cfgS_out.txt --cmd "$C" --out $O > /dev/null ;;
    hbm)
      for m in uniform asgen reorder; do
        H="python tools/rmat_pmc_case.py $m"
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${R}_hbm_$m -- $H > /dev/null 2> $O/prof_${R}_hbm_$m.err
        rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_${R}_hbm_$m -- $H > /dev/null 2> $O/pmcf_${R}_hbm_$m.err
        rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_${R}_hbm_$m -- $H > /dev/null 2> $O/pmcw_${R}_hbm_$m.err
        python tools/summarize_rocprof.py --tag ${R}_hbm_$m --stats $O/prof_${R}_hbm_$m --fetch $O/pmcf_${R}_hbm_$m --write $O/pmcw_${R}_hbm_$m --largest-grid --cmd "$H" --out $O > /dev/null 2> $O/${R}_hbm_sum_$m.err
      done ;;
  esac
done
rm -rf $O/prof_${R}_*/ $O/pmcf_${R}_*/ $O/pmcw_${R}_*/ $O/pmcl_${R}_*/ 2>/dev/null
ls -la $O/${R}_* 2>/dev/null
