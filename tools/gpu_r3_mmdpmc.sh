#!/bin/bash
# MMD chain at the A2GNN shapes on the final kernels: per-kernel durations + PMC counters (separate passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
{
echo "# tools/mmd_bench.py 40 under rocprofv3 --kernel-trace --stats (us per launch)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mmd -- python tools/mmd_bench.py 40 > $O/r3_mmd_out.txt 2> $O/r3_mmd.err
tail -1 $O/r3_mmd_out.txt
python tools/kstats.py $O/prof_mmd k_ 2>&1 | grep -E "k_bwd|k_pairdist|k_rowstats|k_bandwidth|k_finalize" | cut -c1-150
rm -rf $O/prof_mmd
echo "# rocprofv3 --pmc passes (average per launch over 10 calls)"
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rocprofv3 --pmc $pmc --output-format csv -d $O/prof_pmc -- python tools/mmd_bench.py 5 > /dev/null 2> $O/r3_mmd_pmc.err
  python - "$O/prof_pmc" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'k_bwd<' in r['Kernel_Name'] or 'k_pairdist' in r['Kernel_Name']:
        acc[(r['Kernel_Name'].replace('(anonymous namespace)::','')[:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k[0], k[1], round(sum(v)/len(v)))
P
  rm -rf $O/prof_pmc
done
} > $O/r3_mmd_pmc.txt 2>&1
cat $O/r3_mmd_pmc.txt
