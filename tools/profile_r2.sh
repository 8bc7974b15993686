#!/bin/bash
# Round-2 measurement session on the GPU box: bench lines + rocprofv3 kernel traces + separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) for (a) the default cfg-A bench, (b) the 5 M-node / 100 M-edge aggregation
# (tools/spmm_sweep.py --big), (c) the cfg-S sampled bench.  Raw output under gpurun_out/, summaries for profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-probe"
B="python tools/spmm_sweep.py --big"
C="python bench.py --workload cfgS --steps 20 --warmup 5 --no-cpu-baseline"
prof() {   # tag cmd...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_${tag}_out.txt 2> $O/prof_${tag}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -- "$@" > /dev/null 2> $O/pmcf_${tag}.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -- "$@" > /dev/null 2> $O/pmcw_${tag}.err
}
python bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err; tail -c 600 $O/r2_bench_default.err
$C > $O/r2_bench_cfgS.json 2> $O/r2_bench_cfgS.err; tail -c 300 $O/r2_bench_cfgS.err
python bench.py --workload cfgS --steps 20 --warmup 5 > $O/r2_bench_cfgS_cpu.json 2>> $O/r2_bench_cfgS.err
prof r2_cfgA $A
prof r2_spmm5m $B
prof r2_cfgS $C
python tools/summarize_rocprof.py --tag r2_cfgA --stats $O/prof_r2_cfgA --fetch $O/pmcf_r2_cfgA --write $O/pmcw_r2_cfgA --bench $O/prof_r2_cfgA_out.txt --cmd "$A" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r2_spmm5m --stats $O/prof_r2_spmm5m --fetch $O/pmcf_r2_spmm5m --write $O/pmcw_r2_spmm5m --cmd "$B" --out $O > /dev/null
python tools/summarize_rocprof.py --tag r2_cfgS --stats $O/prof_r2_cfgS --fetch $O/pmcf_r2_cfgS --write $O/pmcw_r2_cfgS --bench $O/prof_r2_cfgS_out.txt --cmd "$C" --out $O > /dev/null
cp $O/prof_r2_spmm5m_out.txt $O/r2_spmm_sweep.jsonl
ls -la $O/*_rocprof_summary.md
